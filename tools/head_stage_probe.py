"""Stage by stage: the module-by-module head against coclr_gemm_fused on the query encoder's step-1 operands
(feature map and head parameters after one Adam step) of the CoCLR small case."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _cases import build_model, case_inputs, load_golden, loss_fn
import model.pretrain as product
import coclr_amd.model.pretrain as impl
from coclr_amd import ops

gold = load_golden("coclr_s3d_small")
cfg = gold["cfg"]
model = build_model(cfg, product).cuda().train()
model.sampler.eval()
opt = torch.optim.Adam([{"params": p} for _, p in model.named_parameters()], lr=1e-3, weight_decay=1e-5)
feats = []
real_pool = ops.global_avgpool_fwd
ops.global_avgpool_fwd = lambda x, y: (feats.append(x.clone()) if not torch.cuda.is_current_stream_capturing() else None,
                                       real_pool(x, y))[1]
for step in range(2):
    blocks, extra = case_inputs(cfg, step)
    torch.manual_seed(cfg["perm_seed"] + step)
    out, tgt = model(blocks[0].cuda(), blocks[1].cuda(), extra.cuda())
    if step == 0:
        loss = loss_fn("coclr", out, tgt)
        opt.zero_grad(); loss.backward(); opt.step()
ops.global_avgpool_fwd = real_pool
torch.cuda.synchronize()
for which, feat in (("step 0 query", feats[2]), ("step 1 query", feats[-1])):
    fc1, fc2 = model.encoder_q[2], model.encoder_q[4]
    if which.startswith("step 0"):
        continue
    N_, Cf = feat.shape[:2]
    Ch, D = fc1.weight.shape[0], fc2.weight.shape[0]
    w1, b1, w2, b2 = fc1.weight.reshape(Ch, Cf), fc1.bias, fc2.weight.reshape(D, Ch), fc2.bias
    E = lambda *s: torch.empty(*s, device="cuda")
    h0 = E(N_, Cf); ops.global_avgpool_fwd(feat, h0.view(N_, Cf, 1, 1, 1))
    sp1, sp2 = impl._fc_splits(N_, Ch, Cf), impl._fc_splits(N_, D, Ch)
    # module by module
    y1 = E(N_, Ch); ws = E(max(1, ops.gemm_workspace(N_, Ch, Cf, sp1)))
    ops.gemm(h0, Cf, 1, w1, 1, Cf, y1, Ch, b1, N_, Ch, Cf, splits=sp1, workspace=ws)
    h1 = E(N_, Ch); ops.relu_fwd(y1, h1)
    f = E(N_, D); ws = E(max(1, ops.gemm_workspace(N_, D, Ch, sp2)))
    ops.gemm(h1, Ch, 1, w2, 1, Ch, f, D, b2, N_, D, Ch, splits=sp2, workspace=ws)
    q, inv = E(N_, D), E(N_); ops.l2norm_fwd(f, q, inv)
    # fused
    h1f = E(N_, Ch); ws = E(ops.gemm_fused_workspace(N_, Ch, Cf, sp1))
    ops.gemm_fused(h0, Cf, 1, w1, 1, Cf, h1f, Ch, b1, N_, Ch, Cf, relu=True, splits=sp1, workspace=ws)
    ff = E(N_, D); ws = E(ops.gemm_fused_workspace(N_, D, Ch, sp2))
    ops.gemm_fused(h1f, Ch, 1, w2, 1, Ch, ff, D, b2, N_, D, Ch, splits=sp2, workspace=ws)
    qf, invf = E(N_, D), E(N_); ws = E(ops.gemm_fused_workspace(N_, D, Ch, sp2))
    ops.gemm_fused(h1f, Ch, 1, w2, 1, Ch, qf, D, b2, N_, D, Ch, splits=sp2, workspace=ws, mode=2, out2=invf, f=1e-12)
    q2, inv2 = E(N_, D), E(N_); ops.l2norm_fwd(ff, q2, inv2)
    torch.cuda.synchronize()
    d = lambda a, b: (bool(torch.equal(a, b)), float((a - b).abs().max()))
    print(which, "splits", sp1, sp2)
    print("  h1  fused vs gemm+relu            ", d(h1f, h1))
    print("  f   fused(mode 0) vs gemm         ", d(ff, f))
    print("  q   l2norm(f fused) vs l2norm(f)  ", d(q2, q))
    print("  q   fused(mode 2) vs l2norm(f)    ", d(qf, q), " inv", d(invf, inv))
    print("  sum of squares per row", (f * f).sum(1).tolist(), "inv", inv.tolist())
