"""Cost of the accumulate epilogue (dX += ...) of the data-gradient kernels at the benchmark shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import ops, engine
B = 32
dev = torch.device("cuda")
run = engine.Run(dev, False)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
L = [("3b.group", 192, 176, (1,1,1), (0,0,0), (16,16,16)),
     ("3c.group", 256, 288, (1,1,1), (0,0,0), (16,16,16)),
     ("4b.group", 480, 304, (1,1,1), (0,0,0), (8,8,8)),
     ("4f.group", 528, 448, (1,1,1), (0,0,0), (8,8,8)),
     ("5c.group", 832, 624, (1,1,1), (0,0,0), (4,4,4)),
     ("3c.b1.conv1", 128, 192, (1,3,3), (0,1,1), (16,16,16)),
     ("4f.b1.conv1", 160, 320, (1,3,3), (0,1,1), (8,8,8)),
     ("3c.b1.conv2", 192, 192, (3,1,1), (1,0,0), (16,16,16))]
print("%-14s | dgrad ms  plain   accumulate" % "layer")
for name, cin, cout, k, p, idim in L:
    g = ops.conv_geom(B, cin, cout, idim, k, (1,1,1), p)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    wpt = run.pack(w, True, algo=g.dgrad().algo)
    dy = torch.randn(B, cout, *g.odim, device=dev); dx = torch.zeros(B, cin, *idim, device=dev)
    t0 = timeit(lambda: ops.conv_fwd(g.dgrad(), dy, wpt, dx))
    t1 = timeit(lambda: ops.conv_fwd(g.dgrad(), dy, wpt, dx, accumulate=True))
    print("%-14s | %16.3f %12.3f" % (name, t0, t1))
