"""Is the query encoder's forward a function of its explicit inputs only?  The same training forward
(identical parameters and clip) repeated with the caching allocator's free blocks poisoned in between
(NaN, then 1e30, then a different allocation history): q must be bit-identical every time.  A difference
means some kernel reads memory it did not write in this pass, or picks a variant by address."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _cases import build_model, case_inputs, load_golden
import model.pretrain as product
import coclr_amd.model.pretrain as impl
from coclr_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "coclr"
gold = load_golden("%s_s3d_small" % kind)
cfg = gold["cfg"]
model = build_model(cfg, product).cuda().train()
if kind == "coclr":
    model.sampler.eval()
rec = []
real = ops.nce_logits_fwd


def spy(q, k, queue, logits, T):
    rec.append(q.clone())
    return real(q, k, queue, logits, T)


ops.nce_logits_fwd = spy
blocks, extra = case_inputs(cfg, 1)
args = [blocks[0].cuda()] if kind == "infonce" else [blocks[0].cuda(), blocks[1].cuda(), extra.cuda()]


def poison(val, sizes):
    junk = [torch.full((n,), val, device="cuda") for n in sizes]
    torch.cuda.synchronize()
    del junk


def run(tag):
    torch.manual_seed(5)
    out = model(*args)
    torch.cuda.synchronize()
    del out
    print(tag, "q identical to the first pass:", bool(torch.equal(rec[-1], rec[0])),
          float((rec[-1] - rec[0]).abs().max()), "finite:", bool(torch.isfinite(rec[-1]).all()))


run("first")
run("repeat")
sizes = [1 << k for k in range(8, 25)] + [3 * (1 << k) + 17 for k in range(8, 24)]
torch.cuda.empty_cache()
poison(float("nan"), sizes * 3)
run("after NaN poisoning")
torch.cuda.empty_cache()
run("after empty_cache")
poison(1e30, sizes * 3)
run("after 1e30 poisoning")
for fused in (False, True):
    impl.FUSED_HEAD = fused
    run("FUSED_HEAD=%s" % fused)
    run("FUSED_HEAD=%s again" % fused)
