"""CPU-oracle step time vs intra-op thread count on this host (config 1: B=4, K=2048)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import coclr_oracle as orc
from model.pretrain import InfoNCE
B, K = 4, 2048
torch.manual_seed(0)
model = InfoNCE("s3d", 128, K, 0.999, 0.07)
for nt in [int(a) for a in sys.argv[1:]]:
    torch.set_num_threads(nt)
    sd = orc.training_state(model.state_dict())
    leaves = [sd[k] for k, _ in model.named_parameters() if sd[k].requires_grad]
    opt = torch.optim.Adam([{"params": p} for p in leaves], lr=1e-3, weight_decay=1e-5)
    ts = []
    for step in range(3):
        g = torch.Generator().manual_seed(100 + step)
        block = torch.randn(B, 2, 3, 32, 128, 128, generator=g)
        perm = torch.randperm(B, generator=g)
        t0 = time.perf_counter()
        opt.zero_grad()
        (logits, labels), = orc.nce_step(sd, "infonce", "s3d", [block], None, 128, K, 0.999, 0.07, perm)
        F.cross_entropy(logits, labels).backward()
        opt.step()
        if step: ts.append(time.perf_counter() - t0)
    print("threads %3d: %.2f s/step  %.2f clips/s" % (nt, sum(ts) / len(ts), B / (sum(ts) / len(ts))), flush=True)
