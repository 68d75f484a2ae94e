"""Weight-gradient workgroup order, A/B per geometry on one box: every distinct weight-gradient launch of
one S3D InfoNCE training step at B=32 is captured (ops.conv_wgrad arguments), then replayed alone with
COCLR_WGRAD_ORDER=split and =tile, alternating.  GPU box only.
usage: python tools/wgrad_order_ab.py [rounds]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from coclr_amd import ops
from model.pretrain import InfoNCE

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(os.environ.get("B", "32"))
dev = torch.device("cuda")
seen = {}
inner = ops.conv_wgrad


def spy(geom, x, dy, dw, ws, co_stride, ci_stride, tap_base, accumulate=False):
    multi = isinstance(dw, (list, tuple))
    key = (geom.Cin, geom.Cout, geom.idim, geom.k, geom.s, geom.p, geom.algo, tap_base,
           tuple(t.shape[0] for t in dw) if multi else None)
    if key not in seen:
        seen[key] = (geom, tuple(x.shape), tuple(dy.shape), [tuple(t.shape) for t in dw] if multi else tuple(dw.shape),
                     ws.numel(), co_stride, ci_stride, tap_base)
    return inner(geom, x, dy, dw, ws, co_stride, ci_stride, tap_base, accumulate)


ops.conv_wgrad = spy
from coclr_amd import engine
engine.ops.conv_wgrad = spy
torch.manual_seed(0)
model = InfoNCE('s3d', 128, 2048, 0.999, 0.07).cuda().train()
block = torch.randn(B, 2, 3, 32, 128, 128, device=dev)
out, tgt = model(block)
F.cross_entropy(out, tgt).backward()
torch.cuda.synchronize()
del model, block, out
torch.cuda.empty_cache()
ops.conv_wgrad = inner


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("%-44s %6s | %9s %9s | %6s" % ("geometry (Cin->Cout k in)", "splits", "split ms", "tile ms", "tile/split"))
tot = [0.0, 0.0]
for key, (geom, xs, dys, dws, wsn, cs, cis, tb) in sorted(seen.items(), key=lambda kv: str(kv[0])):
    x = torch.randn(xs, device=dev)
    dy = torch.randn(dys, device=dev)
    dw = [torch.empty(s, device=dev) for s in dws] if isinstance(dws, list) else torch.empty(dws, device=dev)
    ws = torch.empty(wsn, device=dev)
    t = {"split": [], "tile": []}
    for _ in range(rounds):
        for order in ("split", "tile"):
            os.environ["COCLR_WGRAD_ORDER"] = order
            t[order].append(timeit(lambda: inner(geom, x, dy, dw, ws, cs, cis, tb)))
    a, b = min(t["split"]), min(t["tile"])
    tot[0] += a
    tot[1] += b
    splits = wsn // (geom.Cout * geom.Cin * geom.k[0] * geom.k[1] * geom.k[2])
    print("%4d->%-4d %-9s %-14s a%d t%-3d %-6s %6d | %9.4f %9.4f | %6.3f" % (
        geom.Cin, geom.Cout, "x".join(map(str, geom.k)), "x".join(map(str, geom.idim)), geom.algo, tb,
        "multi" if isinstance(dws, list) else "", splits, a, b, b / a))
    del x, dy, dw, ws
print("sum over distinct launches: split %.3f ms, tile %.3f ms" % tuple(tot))
