"""Which host-side ops launch the small ATen / runtime kernels inside one training step
(copyBuffer, fill, mul ...)?  torch.profiler with stacks, one step, grouped by op + caller."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.distributed as dist
from torch.profiler import profile, ProfilerActivity

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from model.pretrain import InfoNCE
from coclr_amd import loss as L
B = int(os.environ.get("B", "32"))
torch.manual_seed(0)
model = InfoNCE("s3d", 128, 2048, 0.999, 0.07).cuda()
ddp = nn.parallel.DistributedDataParallel(model, device_ids=[0])
opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3, weight_decay=1e-5)
crit = L.CrossEntropyLoss()
x = torch.randn(B, 2, 3, 32, 128, 128, device="cuda")
def step():
    out, tgt = ddp(x)
    loss = crit(out, tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
# device-side kernels by name
kern = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
print("== device kernels launched by ATen / runtime (count per step) ==")
for k, n in kern.most_common():
    if "anonymous namespace" in k and "at::native" not in k:
        continue
    print("%5d  %s" % (n, k))
print("== host ops (aten::*, Memcpy, Memset) with their python caller ==")
ops = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU:
        continue
    nm = e.name
    if not (nm.startswith("aten::") or "emcpy" in nm or "emset" in nm):
        continue
    if nm in ("aten::empty", "aten::view", "aten::as_strided", "aten::empty_like", "aten::empty_strided",
              "aten::select", "aten::slice", "aten::reshape", "aten::detach", "aten::alias", "aten::t",
              "aten::transpose", "aten::expand", "aten::unsqueeze", "aten::squeeze", "aten::narrow",
              "aten::_unsafe_view", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense",
              "aten::result_type", "aten::lift_fresh", "aten::resolve_conj", "aten::resolve_neg",
              "aten::contiguous", "aten::view_as", "aten::unbind", "aten::to", "aten::set_"):
        continue
    stack = [f for f in (e.stack or []) if "site-packages/torch/" not in f and "dist-packages/torch/" not in f]
    top = stack[0].strip()[-90:] if stack else "<no python frame: autograd engine / DDP reducer (C++)>"
    ops[(nm, top)] += 1
for (nm, top), n in ops.most_common(45):
    print("%5d  %-28s %s" % (n, nm, top))
dist.destroy_process_group()
