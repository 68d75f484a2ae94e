"""GPU time per phase of the training step (HIP events on the main stream, no profiler)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from model.pretrain import InfoNCE
import bench
torch.manual_seed(0)
model = InfoNCE("s3d", 128, 2048, 0.999, 0.07).cuda()
ddp = nn.parallel.DistributedDataParallel(model, device_ids=[0])
opt = torch.optim.Adam([p for p in ddp.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5, fused=True)
crit = nn.CrossEntropyLoss().cuda()
ddp.train()
blocks = [bench.synthetic_block(32, 32, 128, torch.device("cuda"), 1234 + i) for i in range(2)]
names = ["forward", "loss", "backward(+ddp hooks)", "adam"]
acc = [0.0] * 4
N = 15
for it in range(5 + N):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    ev[0].record()
    out, tgt = ddp(blocks[it % 2]); ev[1].record()
    loss = crit(out, tgt); ev[2].record()
    opt.zero_grad(set_to_none=True)
    loss.backward(); ev[3].record()
    opt.step(); ev[4].record()
    if it >= 5:
        torch.cuda.synchronize()
        for i in range(4):
            acc[i] += ev[i].elapsed_time(ev[i + 1])
print("GPU ms per phase (main stream):", {n: round(a / N, 2) for n, a in zip(names, acc)}, "sum %.2f" % (sum(acc) / N))
dist.destroy_process_group()
