cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
python - <<'PY'
import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
from model.pretrain import InfoNCE
from coclr_amd import engine
torch.manual_seed(0)
m = InfoNCE('s3d', 128, 2048, 0.999, 0.07).cuda()
ddp = torch.nn.parallel.DistributedDataParallel(m, device_ids=[0])
opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5)
crit = torch.nn.CrossEntropyLoss().cuda()
ddp.train()
g = torch.Generator(device="cuda").manual_seed(1)
mem = []
t0 = time.time()
for step in range(300):
    x = torch.randn(32, 2, 3, 32, 128, 128, device="cuda", generator=g)      # a NEW input tensor every step
    out, tgt = ddp(x)
    loss = crit(out, tgt)
    opt.zero_grad(); loss.backward(); opt.step()
    if step % 50 == 49:
        torch.cuda.synchronize()
        mem.append((step + 1, round(torch.cuda.memory_allocated() / 2**30, 2), round(torch.cuda.memory_reserved() / 2**30, 2), round(float(loss), 4)))
print("steps, allocated GiB, reserved GiB, loss:", mem)
print("plans", engine.PLAN_STATS, "time %.1f s" % (time.time() - t0))
assert all(l == l for *_, l in mem)
assert mem[-1][2] <= mem[1][2] + 0.5, "reserved memory grows"
PY
