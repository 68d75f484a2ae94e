"""Is the S3D forward bit-reproducible stage by stage at a given clip shape?  Runs every stage twice on
the same input (train-mode BatchNorm, no grad) and against float64 ATen on the host.  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from backbone.select_backbone import select_backbone
T = int(os.environ.get("T", "8")); HW = int(os.environ.get("HW", "64")); B = int(os.environ.get("B", "4"))
torch.manual_seed(0)
net, _ = select_backbone("s3d")
# He-normal weights as in the drop-in fixtures
g = torch.Generator().manual_seed(501)
with torch.no_grad():
    for k, v in net.state_dict().items():
        if v.dim() == 5:
            v.copy_(torch.randn(v.shape, generator=g) * (2.0 / (v.shape[1] * v.shape[2] * v.shape[3] * v.shape[4])) ** 0.5)
net = net.cuda().train()
x = torch.randn(B, 3, T, HW, HW, generator=g).cuda()
with torch.no_grad():
    for name in ("block1", "block2", "block3", "block4", "block5"):
        blk = getattr(net, name)
        mods = list(blk)
        cur = x
        for i, m in enumerate(mods):
            outs = [m(cur).clone() for _ in range(3)]
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            d = max(float((outs[0] - o).abs().max()) for o in outs[1:])
            print("%-8s %-14s out %-22s reproducible=%s max diff %.3e  |out| max %.3e nan=%s" % (
                name, type(m).__name__ + str(i), tuple(outs[0].shape), same, d, float(outs[0].abs().max()),
                bool(torch.isnan(outs[0]).any())))
            cur = outs[0]
        x = cur
