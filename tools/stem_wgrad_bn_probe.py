"""Conv_1a.conv1 / bn1 backward at the benchmarked size (B = 32), alone on the chip: the two-pass form (BatchNorm
backward reduce + apply, then the stem weight gradient) against the short form (reduce + coefficients, then the weight
gradient that applies them while it loads).  HIP events, 20 repetitions after 3 warm-up."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coclr_amd import ops

N, Cin, Cout, dims, k, s, p = 32, 3, 64, (32, 128, 128), (1, 7, 7), (1, 2, 2), (0, 3, 3)
g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p)
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(N, Cin, *dims, device=dev)
y = torch.randn(N, Cout, *g.odim, device=dev)
dz = torch.randn_like(y)
small = torch.rand(4, Cout, device=dev) + 0.5
small[0] -= 1.0
sums = torch.empty(ops.bn_backward_workspace(N, Cout), dtype=torch.float64, device=dev)
dgb = torch.empty(2, Cout, device=dev)
coef = torch.empty(5, Cout, device=dev)
dy = torch.empty_like(y)
dw = torch.empty(Cout, Cin, *k, device=dev)
ws = torch.empty(g.wgrad_workspace(), device=dev)
kk = 49


def two_pass():
    ops.bn_act_backward(dz, y, None, small[2], small[3], small[0], small[1], sums, dy, None, dgb[0], dgb[1], True, True)
    ops.conv_wgrad(g, x, dy, dw, ws, Cin * kk, kk, 0)


def short():
    ops.bn_act_backward_coeffs(dz, y, small[2], small[3], small[0], small[1], sums, coef, dgb[0], dgb[1], True, True)
    ops.conv_wgrad_bn(g, x, dz, y, coef, True, dw, ws, Cin * kk, kk)


def wgrad_only():
    ops.conv_wgrad(g, x, dy, dw, ws, Cin * kk, kk, 0)


def wgrad_bn_only():
    ops.conv_wgrad_bn(g, x, dz, y, coef, True, dw, ws, Cin * kk, kk)


for name, fn in (("two-pass (reduce, apply, wgrad)", two_pass), ("short (reduce+coeffs, wgrad_bn)", short),
                 ("stem wgrad alone", wgrad_only), ("stem wgrad_bn alone", wgrad_bn_only)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-34s %.3f ms" % (name, e0.elapsed_time(e1) / 20))
