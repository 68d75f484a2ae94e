"""Per-call timing of every kernel-library call in one S3D encoder forward+backward at the
benchmark shape (B=32, 3x32x128x128), grouped by (op, geometry).  GPU box only."""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import ops

B = int(os.environ.get("B", "32"))
REPS = 3
records = []
enabled = [False]


def wrap(name):
    inner = getattr(ops, name)

    def f(*a, **kw):
        if not enabled[0]:
            return inner(*a, **kw)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); r = inner(*a, **kw); e1.record()
        g = a[0]
        if isinstance(g, ops.ConvGeom):
            key = "%d->%d k%s s%s d%s in%s%s" % (g.Cin, g.Cout, "x".join(map(str, g.k)), "x".join(map(str, g.s)),
                                              "x".join(map(str, g.d)), "x".join(map(str, g.idim)),
                                              (" lat" if g.lattice else "") + (" wino" if g.algo else ""))
            flops = 2.0 * g.N * g.Cin * g.Cout * g.taps * g.odim[0] * g.odim[1] * g.odim[2]
            if g.d != (1, 1, 1):   # dilated dgrad: useful flops are those of the forward conv
                flops /= (g.d[0] * g.d[1] * g.d[2])
        elif isinstance(g, ops.PoolGeom):
            key = "C%d k%s s%s in%s" % (g.C, "x".join(map(str, g.k)), "x".join(map(str, g.s)), "x".join(map(str, g.idim)))
            flops = 0.0
        else:
            t = next((x for x in a if torch.is_tensor(x)), None)
            key = "x".join(map(str, t.shape)) if t is not None else ""
            flops = 0.0
        records.append((name, key, flops, e0, e1))
        return r
    setattr(ops, name, f)


for n in ("conv_fwd", "conv_wgrad", "conv_pack_weights", "bn_finalize", "bn_act_apply", "bn_act_backward",
          "maxpool_fwd", "maxpool_bwd"):
    wrap(n)

from backbone.select_backbone import select_backbone
torch.manual_seed(0)
net, _ = select_backbone("s3d")
net = net.cuda().train()
x = torch.randn(B, 3, 32, 128, 128, device="cuda")
for it in range(1 + REPS):
    enabled[0] = it > 0
    y = net(x)
    y.backward(torch.randn_like(y))
    net.zero_grad(set_to_none=True)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, key, flops, e0, e1 in records:
    a = agg.setdefault((name, key), [0, 0.0, 0.0])
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += flops
tot = sum(v[1] for v in agg.values()) / REPS
print("one S3D encoder fwd+bwd at B=%d: %.2f ms in library calls (event-timed, includes launch gaps)" % (B, tot))
byop = collections.Counter()
for (name, key), (n, ms, fl) in agg.items():
    byop[name] += ms / REPS
for k, v in byop.most_common():
    print("  %-18s %7.2f ms" % (k, v))
print("%-12s %-62s %5s %9s %7s" % ("op", "geometry", "calls", "ms/iter", "TF/s"))
for (name, key), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if not name.startswith("conv_") or name == "conv_pack_weights":
        continue
    print("%-12s %-62s %5d %9.3f %7.1f" % (name, key, n // REPS, ms / REPS, fl / ms / 1e9 if ms else 0))
