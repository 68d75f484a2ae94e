"""In-step marginal cost of a kernel family: bench.py with the family's launches SKIPPED (wrong results by design --
a timing ablation, never a measurement of the product).  usage: COCLR_ABLATE=<family> python tools/ablate_step.py
[bench args]; families: bn_apply, bn_bwd, pool, wgrad, dgrad, bn_multi, none."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coclr_amd import ops  # noqa: E402

fam = os.environ.get("COCLR_ABLATE", "none")


def skip(*a, **kw):
    return None


if fam == "bn_apply":
    ops.bn_act_apply = skip
elif fam == "bn_bwd":
    ops.bn_act_backward = skip
    ops.bn_act_backward_pooled = skip
elif fam == "bn_multi":
    ops.bn_finalize_apply_multi = skip
    ops.bn_act_backward_multi = skip
elif fam == "pool":
    ops.maxpool_fwd = skip
    ops.maxpool_bwd = skip
elif fam == "wgrad":
    ops.conv_wgrad = skip
elif fam == "dgrad":
    # data gradients are conv_fwd calls on geometries made by ConvGeom.dgrad() / dgrad_phases() (a destination lattice)
    real_fwd, real_multi, orig_dgrad = ops.conv_fwd, ops.conv_fwd_multi, ops.ConvGeom.dgrad
    tags = set()

    def tagged(self):
        g = orig_dgrad(self)
        tags.add(id(g))
        return g
    ops.ConvGeom.dgrad = tagged
    is_dgrad = lambda g: id(g) in tags or g.lattice is not None
    ops.conv_fwd = lambda g, *a, **kw: None if is_dgrad(g) else real_fwd(g, *a, **kw)

    def multi(calls):
        keep = [c for c in calls if not is_dgrad(c["geom"])]
        if keep:
            real_multi(keep)
    ops.conv_fwd_multi = multi
import bench  # noqa: E402
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
bench.main()
