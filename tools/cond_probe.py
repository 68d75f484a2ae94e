"""Per-tensor gradient error of the product on the de-saturated fixture (tests/golden/
infonce_s3d_conditioned.pt) against the reference's own float64 gradients; the reference's fp32
error beside it.  GPU box only.  Env switches (COCLR_WINOGRAD=0 ...) select kernel families."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _cases import *
import model.pretrain as product
gold = load_golden("infonce_s3d_conditioned")
cfg = gold["cfg"]; rec = gold["steps"][0]
model = build_model(cfg, product).cuda()
model.train()
blocks, extra = case_inputs(cfg, 0)
torch.manual_seed(cfg["perm_seed"])
out, tgt = model(blocks[0].cuda())
loss = loss_fn("infonce", out, tgt)
loss.backward()
torch.cuda.synchronize()
grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
truth = recorded_truth(rec)
print("logits err %.2e  loss %.6f ref %.6f fp64 %.6f" % (rel_err(out, rec["logits"]), float(loss.detach()), float(rec["loss"]), float(rec["loss64"])))
for k, ref in rec["grads"].items():
    t = truth[k]
    g = sample(grads[k])
    print("%-52s max: ref %.2e got %.2e | L2: ref %.2e got %.2e" % (k[10:], rel_err(ref, t), rel_err(g, t), l2_err(ref, t), l2_err(g, t)))
