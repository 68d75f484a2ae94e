"""Per-layer kernel timing at the benchmark shapes (B=32, 3x32x128x128 S3D): forward conv,
data gradient, weight gradient, pools.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import ops, engine

B = int(os.environ.get("B", "32"))
dev = torch.device("cuda")
run = engine.Run(dev, False)

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

# (name, Cin, Cout, k, s, p, in dims)
L = [("Conv_1a.conv1", 3, 64, (1,7,7), (1,2,2), (0,3,3), (32,128,128)),
     ("Conv_1a.conv2", 64, 64, (7,1,1), (2,1,1), (3,0,0), (32,64,64)),
     ("Conv_2b", 64, 64, (1,1,1), (1,1,1), (0,0,0), (16,32,32)),
     ("Conv_2c.conv1", 64, 192, (1,3,3), (1,1,1), (0,1,1), (16,32,32)),
     ("Conv_2c.conv2", 192, 192, (3,1,1), (1,1,1), (1,0,0), (16,32,32)),
     ("3b.b0", 192, 64, (1,1,1), (1,1,1), (0,0,0), (16,16,16)),
     ("3b.b1a", 192, 96, (1,1,1), (1,1,1), (0,0,0), (16,16,16)),
     ("3b.b1.conv1", 96, 128, (1,3,3), (1,1,1), (0,1,1), (16,16,16)),
     ("3b.b1.conv2", 128, 128, (3,1,1), (1,1,1), (1,0,0), (16,16,16)),
     ("3b.b2a", 192, 16, (1,1,1), (1,1,1), (0,0,0), (16,16,16)),
     ("3b.b2.conv1", 16, 32, (1,3,3), (1,1,1), (0,1,1), (16,16,16)),
     ("3c.b1.conv1", 128, 192, (1,3,3), (1,1,1), (0,1,1), (16,16,16)),
     ("3c.b1.conv2", 192, 192, (3,1,1), (1,1,1), (1,0,0), (16,16,16)),
     ("3c.b0", 256, 128, (1,1,1), (1,1,1), (0,0,0), (16,16,16)),
     ("4b.b0", 480, 192, (1,1,1), (1,1,1), (0,0,0), (8,8,8)),
     ("4f.b1.conv1", 160, 320, (1,3,3), (1,1,1), (0,1,1), (8,8,8)),
     ("3c.group", 256, 288, (1,1,1), (1,1,1), (0,0,0), (16,16,16)),
     ("4b.group", 480, 304, (1,1,1), (1,1,1), (0,0,0), (8,8,8)),
     ("4b.b2.conv1", 16, 48, (1,3,3), (1,1,1), (0,1,1), (8,8,8)),
     ("4c.b1.conv1", 112, 224, (1,3,3), (1,1,1), (0,1,1), (8,8,8)),
     ("4f.b1.conv2", 320, 320, (3,1,1), (1,1,1), (1,0,0), (8,8,8)),
     ("4f.b0", 528, 256, (1,1,1), (1,1,1), (0,0,0), (8,8,8)),
     ("5c.b0", 832, 384, (1,1,1), (1,1,1), (0,0,0), (4,4,4)),
     ("5c.b1.conv1", 192, 384, (1,3,3), (1,1,1), (0,1,1), (4,4,4)),
     ("5c.b1.conv2", 384, 384, (3,1,1), (1,1,1), (1,0,0), (4,4,4))]
only = sys.argv[1:] 
print("%-16s %9s | %8s %7s | %8s %7s | %8s %7s" % ("layer", "GF", "fwd ms", "TF/s", "dgrad ms", "TF/s", "wgrad ms", "TF/s"))
for name, cin, cout, k, s, p, idim in L:
    if only and not any(o in name for o in only): continue
    g = ops.conv_geom(B, cin, cout, idim, k, s, p)
    x = torch.randn(B, cin, *idim, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    y = torch.empty(B, cout, *g.odim, device=dev)
    stats = torch.empty(2 * cout * g.ntiles(), device=dev)
    wp = run.pack(w, False, algo=g.algo); wpt = run.pack(w, True, algo=g.dgrad().algo)
    gf = 2.0 * B * cout * cin * k[0]*k[1]*k[2] * g.odim[0]*g.odim[1]*g.odim[2] / 1e9
    tf = timeit(lambda: ops.conv_fwd(g, x, wp, y, stats=stats))
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    td = timeit(lambda: ops.conv_fwd(g.dgrad(), dy, wpt, dx)) if cin > 3 else float('nan')
    dw = torch.empty_like(w); ws = torch.empty(g.wgrad_workspace(), device=dev); kk = k[0]*k[1]*k[2]
    tw = timeit(lambda: ops.conv_wgrad(g, x, dy, dw, ws, cin*kk, kk, 0))
    print("%-16s %9.2f | %8.3f %7.1f | %8.3f %7.1f | %8.3f %7.1f" % (name, gf, tf, gf/tf, td, gf/td, tw, gf/tw))
    del x, y, dy, dx, ws

print()
P = [("MaxPool_2a", 64, (1,3,3), (1,2,2), (0,1,1), (16,64,64)),
     ("MaxPool_3a", 192, (1,3,3), (1,2,2), (0,1,1), (16,32,32)),
     ("3b.pool", 192, (3,3,3), (1,1,1), (1,1,1), (16,16,16)),
     ("MaxPool_4a", 480, (3,3,3), (2,2,2), (1,1,1), (16,16,16)),
     ("4b.pool", 480, (3,3,3), (1,1,1), (1,1,1), (8,8,8)),
     ("MaxPool_5a", 832, (2,2,2), (2,2,2), (0,0,0), (8,8,8)),
     ("5b.pool", 832, (3,3,3), (1,1,1), (1,1,1), (4,4,4))]
print("%-12s %9s | %8s %8s | %8s %8s" % ("pool", "MB in+out", "fwd ms", "GB/s", "bwd ms", "GB/s"))
for name, c, k, s, p, idim in P:
    if only and not any(o in name for o in only): continue
    g = ops.PoolGeom(B, c, idim, k, s, p)
    x = torch.relu(torch.randn(B, c, *idim, device=dev))
    y = torch.empty(B, c, *g.odim, device=dev); idx = torch.empty(B, c, *g.odim, dtype=torch.int32, device=dev)
    tf = timeit(lambda: ops.maxpool_fwd(g, x, y, idx))
    dy = torch.randn_like(y); dx = torch.empty_like(x)
    tb = timeit(lambda: ops.maxpool_bwd(g, dy, idx, dx))
    mbf = (x.numel() + 2 * y.numel()) * 4 / 1e6
    mbb = (x.numel() + 2 * y.numel()) * 4 / 1e6
    print("%-12s %9.1f | %8.3f %8.0f | %8.3f %8.0f" % (name, mbf, tf, mbf/tf, tb, mbb/tb))

# ---- BatchNorm (HBM-bound): apply = 2|y| bytes, backward = 5|y| bytes -----------------------
print()
print("%-16s %9s | %8s %8s | %8s %8s" % ("bn unit", "|y| MB", "apply ms", "GB/s", "bwd ms", "GB/s"))
for name, c, idim in [("Conv_1a.bn1", 64, (32, 64, 64)), ("Conv_1a.bn2", 64, (16, 64, 64)),
                      ("Conv_2c.bn1", 192, (16, 32, 32)), ("Mixed_3c.out", 480, (16, 16, 16)),
                      ("Mixed_4f.b1", 320, (8, 8, 8)), ("Mixed_5c.b1", 384, (4, 4, 4))]:
    if only and not any(o in name for o in only): continue
    y = torch.randn(B, c, *idim, device=dev)
    z = torch.empty_like(y); dz = torch.randn_like(y); dy = torch.empty_like(y)
    small = torch.rand(4, c, device=dev) + 0.5
    dgb = torch.empty(2, c, device=dev)
    sums = torch.empty(ops.bn_backward_workspace(B, c), dtype=torch.float64, device=dev)
    ta = timeit(lambda: ops.bn_act_apply(y, small[2], small[3], None, z, True))
    tb = timeit(lambda: ops.bn_act_backward(dz, y, None, small[2], small[3], small[0], small[1], sums, dy,
                                            None, dgb[0], dgb[1], True, True))
    mb = y.numel() * 4 / 1e6
    print("%-16s %9.1f | %8.3f %8.0f | %8.3f %8.0f" % (name, mb, ta, 2 * mb / ta, tb, 5 * mb / tb))

# ---- fused pool + BatchNorm backward of the lazily pooled units ---------------------------------
print()
print("%-16s %9s | %8s" % ("pooled bn bwd", "|y| MB", "ms"))
for name, c, idim, k, s, p in [("Conv_1a.bn2", 64, (16, 64, 64), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                               ("Conv_2c.bn2", 192, (16, 32, 32), (1, 3, 3), (1, 2, 2), (0, 1, 1))]:
    if only and not any(o in name for o in only): continue
    g = ops.PoolGeom(B, c, idim, k, s, p)
    y = torch.randn(B, c, *idim, device=dev)
    small = torch.rand(4, c, device=dev) + 0.5
    py = torch.empty(B, c, *g.odim, device=dev); idx = torch.empty(B, c, *g.odim, dtype=torch.int32, device=dev)
    ops.maxpool_fwd(g, y, py, idx, in_scale=small[2], in_shift=small[3], in_relu=True)
    pdy = torch.randn_like(py); dy = torch.empty_like(y); dgb = torch.empty(2, c, device=dev)
    sums = torch.empty(ops.bn_backward_workspace(B, c), dtype=torch.float64, device=dev)
    t = timeit(lambda: ops.bn_act_backward_pooled(g, pdy, idx, y, small[2], small[3], small[0], small[1],
                                                  sums, dy, dgb[0], dgb[1], True, True))
    print("%-16s %9.1f | %8.3f" % (name, y.numel() * 4 / 1e6, t))
