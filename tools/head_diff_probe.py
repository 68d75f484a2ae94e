"""Where do the fused and the module-by-module head first differ (CoCLR small case)?  Records the operands
of every logits launch (q, k, queue) and the mined keys of both runs and compares them step by step."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _cases import build_model, case_inputs, load_golden, loss_fn
import model.pretrain as product
import coclr_amd.model.pretrain as impl
from coclr_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "coclr"
gold = load_golden("%s_s3d_small" % kind)
cfg = gold["cfg"]
base = build_model(cfg, product)
runs = []
for fused in (True, False):
    impl.FUSED_HEAD = fused
    rec = []
    real = ops.nce_logits_fwd
    real_mine = ops.mine_positives

    def spy(q, k, queue, logits, T):
        rec.append(("logits", q.clone(), k.clone(), queue.clone()))
        return real(q, k, queue, logits, T)

    def spy_mine(kf, *a, **kw):
        rec.append(("mine", kf.clone()))
        return real_mine(kf, *a, **kw)
    real_pool = ops.global_avgpool_fwd

    def spy_pool(x, y):
        rec.append(("feat", x.clone()))
        return real_pool(x, y)
    ops.global_avgpool_fwd = spy_pool
    real_conv = ops.conv_fwd
    first = [True]

    def spy_conv(g, x, w, y, *a, **kw):
        r = real_conv(g, x, w, y, *a, **kw)
        if g.Cin == 3 and not torch.cuda.is_current_stream_capturing():
            rec.append(("stem", x.clone().float(), w.clone(), y.clone()))
        return r
    ops.conv_fwd = spy_conv
    ops.nce_logits_fwd, ops.mine_positives = spy, spy_mine
    model = copy.deepcopy(base).cuda().train()
    if kind == "coclr":
        model.sampler.eval()
    opt = torch.optim.Adam([{"params": p} for _, p in model.named_parameters()], lr=1e-3, weight_decay=1e-5)
    params = []
    for step in range(3):
        blocks, extra = case_inputs(cfg, step % cfg["steps"])
        torch.manual_seed(cfg["perm_seed"] + step)
        if kind == "infonce":
            out, tgt = model(blocks[0].cuda())
        else:
            out, tgt = model(blocks[0].cuda(), blocks[1].cuda(), extra.cuda())
        loss = loss_fn(kind, out, tgt)
        opt.zero_grad()
        loss.backward()
        rec.append(("grads", {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
        opt.step()
        rec.append(("params", {k: p.detach().clone() for k, p in model.state_dict().items()}))
        rec.append(("out", out.detach().clone()))
    torch.cuda.synchronize()
    ops.nce_logits_fwd, ops.mine_positives = real, real_mine
    ops.global_avgpool_fwd, ops.conv_fwd = real_pool, real_conv
    runs.append(rec)
print([r[0] for r in runs[0]])
print([r[0] for r in runs[1]])
for i, (a, b) in enumerate(zip(*runs)):
    assert a[0] == b[0]
    if a[0] in ("grads", "params"):
        bad = [k for k in a[1] if not torch.equal(a[1][k], b[1][k])]
        print(i, a[0], "differing tensors:", len(bad), bad[:4])
    else:
        names = {"logits": ("q", "k", "queue"), "mine": ("kf",), "out": ("logits",), "feat": ("feat",),
                 "stem": ("x", "w_packed", "y")}[a[0]]
        print(i, a[0], {n: (bool(torch.equal(x, y)), float((x - y).abs().max())) for n, x, y in zip(names, a[1:], b[1:])})
