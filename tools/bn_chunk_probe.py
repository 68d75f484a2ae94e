"""Does a BatchNorm backward whose apply pass re-reads (dz, y) soon after the reduce pass find them in
L2 / the infinity cache?  Whole-tensor two-pass launch vs the same two kernels issued per chunk of
channels (chunk working set = 2 * N*S*4 bytes per channel).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import ops
dev = torch.device("cuda")
B = 32

def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for name, C_, dims in [("Conv_2c.bn1", 192, (16, 32, 32)), ("Mixed_3c.b1.bn1", 192, (16, 16, 16)),
                       ("Conv_1a.bn2", 64, (16, 64, 64)), ("Mixed_3b.out-ish", 256, (16, 16, 16))]:
    y = torch.randn(B, C_, *dims, device=dev); dz = torch.randn_like(y); dy = torch.empty_like(y)
    small = torch.rand(4, C_, device=dev) + 0.5
    dgb = torch.empty(2, C_, device=dev)
    def run(chunk):
        for c0 in range(0, C_, chunk):
            c1 = min(C_, c0 + chunk)
            sums = torch.empty(ops.bn_backward_workspace(B, c1 - c0), dtype=torch.float64, device=dev)
            ops.bn_act_backward(dz[:, c0:c1], y[:, c0:c1], None, small[2, c0:c1], small[3, c0:c1], small[0, c0:c1],
                                small[1, c0:c1], sums, dy[:, c0:c1], None, dgb[0, c0:c1], dgb[1, c0:c1], True, True)
    mb = y.numel() * 4 / 1e6
    per_ch = 2 * y.numel() // C_ * 4 / 1e6
    line = "%-18s |y| %6.0f MB, %5.1f MB (dz+y) per channel:" % (name, mb, per_ch)
    for chunk in (C_, 64, 32, 16, 8, 4):
        if chunk > C_: continue
        t = timeit(lambda: run(chunk))
        line += "  chunk %3d: %.3f ms (%4.0f GB/s alg)" % (chunk, t, 5 * mb / t)
    print(line)
