"""List host<->device synchronisation points inside a training step (torch's sync debug mode) and time
the host side of each phase.  GPU box only."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29580")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import model.pretrain as product
import bench
from coclr_amd import loss as L
torch.manual_seed(0)
model = product.InfoNCE("s3d", 128, 2048, 0.999, 0.07).cuda()
ddp = nn.parallel.DistributedDataParallel(model, device_ids=[0])
opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3, weight_decay=1e-5)
ddp.train()
blocks = [bench.synthetic_block(32, 32, 128, torch.device("cuda"), 1234 + i) for i in range(2)]
crit = L.CrossEntropyLoss()
def step(i):
    t = [time.perf_counter()]
    out, tgt = ddp(blocks[i % 2]); t.append(time.perf_counter())
    loss = crit(out, tgt); a = L.calc_topk_accuracy(out, tgt, (1, 5)); t.append(time.perf_counter())
    opt.zero_grad(set_to_none=True)
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    return [b - a for a, b in zip(t, t[1:])]
for i in range(5): step(i)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step(5)
    torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step:", len(w))
for x in w[:10]:
    print("  ", str(x.message)[:200], "@", x.filename.split("/")[-1], x.lineno)
acc = [0.0] * 4
t0 = time.perf_counter()
for i in range(20):
    d = step(i)
    acc = [a + b for a, b in zip(acc, d)]
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("host ms per step: forward %.2f  loss %.2f  backward %.2f  optimizer %.2f  | loop %.2f, with final sync %.2f"
      % tuple([a / 20 * 1e3 for a in acc] + [t_host / 20 * 1e3, t_all / 20 * 1e3]))
dist.destroy_process_group()
