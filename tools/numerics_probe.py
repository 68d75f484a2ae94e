"""Where does a gradient lose accuracy?  Elementwise and SUMMED errors of the conv data-gradient
kernels (direct / Winograd) and of the BatchNorm backward against float64 on the host, on the
block-2 shapes of tests/test_gpu_engine.py::test_s3d_stages_per_tensor_gradients.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from coclr_amd import ops, engine

torch.manual_seed(0)
dev = torch.device("cuda")
run = engine.Run(dev, False)


def report(name, got, ref):
    got, ref = got.double().cpu(), ref.double()
    d = got - ref
    s = ref.std()
    csum_ref = ref.sum((0, 2, 3, 4))
    csum_err = (d.sum((0, 2, 3, 4))).abs().max() / csum_ref.abs().max()
    print("%-34s elementwise max %.2e  rms %.2e  mean(bias) %.2e  (units of std);  channel-sum err %.2e"
          % (name, float(d.abs().max() / s), float(d.pow(2).mean().sqrt() / s), float(d.mean() / s),
             float(csum_err)))


for (N, Cin, Cout, dims, k, p) in [(4, 192, 192, (8, 16, 16), (3, 1, 1), (1, 0, 0)),
                                   (4, 64, 192, (8, 16, 16), (1, 3, 3), (0, 1, 1)),
                                   (4, 64, 64, (8, 16, 16), (1, 1, 1), (0, 0, 0))]:
    w = torch.randn(Cout, Cin, *k) * (1.5 / (Cin * k[0] * k[1] * k[2]) ** 0.5)
    dy = torch.randn(N, Cout, *dims)
    dy = dy - dy.mean((0, 2, 3, 4), keepdim=True)          # BatchNorm backward output: zero mean
    x64 = torch.zeros(N, Cin, *dims, dtype=torch.float64, requires_grad=True)
    F.conv3d(x64, w.double(), None, 1, p).backward(dy.double())
    ref = x64.grad
    x32 = torch.zeros(N, Cin, *dims, requires_grad=True)
    F.conv3d(x32, w, None, 1, p).backward(dy)
    report("CPU fp32 %s" % (k,), x32.grad, ref)
    for algo in (0, 1):
        if algo == 1 and k == (1, 1, 1):
            continue
        g = ops.ConvGeom(N, Cin, Cout, dims, k, (1, 1, 1), p, algo=algo)
        dg = g.dgrad()
        dx = torch.empty(N, Cin, *dims, device=dev)
        ops.conv_fwd(dg, dy.cuda(), run.pack(w.cuda(), True, algo=dg.algo), dx)
        report("HIP %s algo=%d" % (k, dg.algo), dx, ref)
        # forward too
        y = torch.empty(N, Cout, *dims, device=dev)
        xin = torch.relu(torch.randn(N, Cin, *dims))
        ops.conv_fwd(g, xin.cuda(), run.pack(w.cuda(), False, algo=g.algo), y)
        report("HIP %s algo=%d forward" % (k, g.algo), y, F.conv3d(xin.double(), w.double(), None, 1, p))

# BatchNorm backward
N, C_, dims = 4, 192, (8, 16, 16)
y = torch.randn(N, C_, *dims) * 1.3 + 0.7
gamma, beta = torch.rand(C_) + 0.5, torch.randn(C_) * 0.3
dz = torch.randn(N, C_, *dims)
y64 = y.double().requires_grad_(True)
g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
z64 = F.relu(F.batch_norm(y64, None, None, g64, b64, True, 0.1, 1e-5))
z64.backward(dz.double())
yd = y.cuda()
cnt = N * dims[0] * dims[1] * dims[2]
mean = y.double().mean((0, 2, 3, 4)); var = y.double().var((0, 2, 3, 4), unbiased=False)
invstd = 1 / torch.sqrt(var + 1e-5)
small = torch.stack([mean, invstd, gamma.double() * invstd, beta.double() - mean * gamma.double() * invstd]).float().cuda()
dyg = torch.empty_like(yd); dgb = torch.empty(2, C_, device=dev)
sums = torch.empty(ops.bn_backward_workspace(N, C_), dtype=torch.float64, device=dev)
ops.bn_act_backward(dz.cuda(), yd, None, small[2], small[3], small[0], small[1], sums, dyg, None, dgb[0], dgb[1], True, True)
report("HIP bn backward dy", dyg, y64.grad)
print("   dgamma err %.2e  dbeta err %.2e" % (float((dgb[0].cpu().double() - g64.grad).abs().max() / g64.grad.abs().max()),
                                              float((dgb[1].cpu().double() - b64.grad).abs().max() / b64.grad.abs().max())))
y32 = y.clone().requires_grad_(True)
F.relu(F.batch_norm(y32, None, None, gamma, beta, True, 0.1, 1e-5)).backward(dz)
report("CPU fp32 bn backward dy", y32.grad, y64.grad)
