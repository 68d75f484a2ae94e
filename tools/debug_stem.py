"""Precision of the stem conv (y and BN partial sums) against a float64 evaluation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from coclr_amd import ops, engine
torch.manual_seed(0)
N = 4
x = torch.randn(N, 3, 16, 64, 64)
w = torch.randn(64, 3, 1, 7, 7) * (1.5 / 147 ** 0.5)
ref = F.conv3d(x.double(), w.double(), None, (1, 2, 2), (0, 3, 3))
ref32 = F.conv3d(x, w, None, (1, 2, 2), (0, 3, 3))
g = ops.ConvGeom(N, 3, 64, x.shape[2:], (1, 7, 7), (1, 2, 2), (0, 3, 3))
run = engine.Run(torch.device("cuda"), save=False)
y = torch.empty(N, 64, *g.odim, device="cuda")
st = torch.empty(2 * 64 * g.ntiles(), device="cuda")
ops.conv_fwd(g, x.cuda(), run.pack(w.cuda(), False), y, stats=st)
torch.cuda.synchronize()
yd = y.double().cpu()
print("ntiles", g.ntiles())
print("y   err: gpu %.3e   cpu32 %.3e" % ((yd - ref).abs().max() / ref.abs().max(), (ref32.double() - ref).abs().max() / ref.abs().max()))
s = st.view(2, 64, -1).double().sum(-1).cpu()
cnt = ref.numel() / 64
mean_ref = ref.sum((0, 2, 3, 4)) / cnt
var_ref = (ref ** 2).sum((0, 2, 3, 4)) / cnt - mean_ref ** 2
mean = s[0] / cnt
var = s[1] / cnt - mean ** 2
print("mean err %.3e  var rel err %.3e" % ((mean - mean_ref).abs().max() / mean_ref.abs().max(), ((var - var_ref).abs() / var_ref).max()))
