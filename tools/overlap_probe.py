"""Do the key encoder (hipGraph on its own stream) and the query encoder really run at the same
time?  HIP events around both, no profiler.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import model.pretrain as product
import coclr_amd.model.pretrain as impl
import bench
from coclr_amd import loss as L
torch.manual_seed(0)
model = product.InfoNCE("s3d", 128, 2048, 0.999, 0.07).cuda()
ddp = nn.parallel.DistributedDataParallel(model, device_ids=[0])
opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3, weight_decay=1e-5)
ddp.train()
blocks = [bench.synthetic_block(32, 32, 128, torch.device("cuda"), 1234 + i) for i in range(2)]
ev = {}
def E(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); ev.setdefault(name, []).append(e)
orig_keys = impl.InfoNCE._encode_keys
orig_enc = impl.InfoNCE._encode
def keys(self, x2, pre=None):
    E("k0"); out = orig_keys(self, x2, pre=pre); E("k1"); return out
def enc(self, encoder, x, n_index=None):
    if encoder is self.encoder_q:
        E("q0"); out = orig_enc(self, encoder, x, n_index=n_index); E("q1"); return out
    return orig_enc(self, encoder, x, n_index=n_index)
impl.InfoNCE._encode_keys = keys
impl.InfoNCE._encode = enc
N = 12
for it in range(4 + N):
    if it == 4: ev.clear()
    E("s0")
    out, tgt = ddp(blocks[it % 2]); E("f1")
    loss = L.cross_entropy(out, tgt)
    opt.zero_grad(set_to_none=True)
    loss.backward(); E("b1")
    opt.step(); E("o1")
torch.cuda.synchronize()
def rel(a, b): return sum(x.elapsed_time(y) for x, y in zip(ev[a], ev[b])) / len(ev[a])
print("per step, ms after step start (averages over %d steps):" % N)
for k in ("k0", "k1", "q0", "q1", "f1", "b1", "o1"):
    print("  %-3s %7.2f" % (k, rel("s0", k)))
print("key encoder span %.2f ms, query forward span %.2f ms, forward total %.2f, backward %.2f, optimizer %.2f"
      % (rel("k0", "k1"), rel("q0", "q1"), rel("s0", "f1"), rel("f1", "b1"), rel("b1", "o1")))
dist.destroy_process_group()
