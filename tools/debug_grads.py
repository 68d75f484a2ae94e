"""Debug helper (GPU box): per-parameter gradient error of the product path vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import torch.nn.functional as F
from _cases import build_model, case_inputs, load_golden, loss_fn, rel_err
from oracle import coclr_oracle as orc
import model.pretrain as product

name = sys.argv[1] if len(sys.argv) > 1 else "infonce_s3d_small"
gold = load_golden(name)
cfg = gold["cfg"]; kind = cfg["kind"]
model = build_model(cfg, product)
sd = orc.training_state(model.state_dict())
model = model.cuda().train()
if kind == "coclr":
    model.sampler.eval()
blocks, extra = case_inputs(cfg, 0)
rec = gold["steps"][0]
torch.manual_seed(cfg["perm_seed"])
args = [b.cuda() for b in blocks] + ([extra.cuda()] if extra is not None else [])
out, tgt = model(*args)
loss = loss_fn(kind, out, tgt)
loss.backward()
pb = [blocks[0]] if kind != "coclr" else [(blocks[0], blocks[1])]
outs = orc.nce_step(sd, kind, cfg["network"], pb, [extra], cfg["dim"], cfg["K"], cfg["m"], cfg["T"],
                    rec["perm"], topk=cfg.get("topk", 5), reverse=cfg.get("reverse", False),
                    sampler_training=False)
rl = loss_fn(kind, *outs[0])
rl.backward()
print("logits err", rel_err(out, outs[0][0]), "loss", float(loss), float(rl))
for k, p in model.named_parameters():
    if p.grad is None:
        continue
    e = rel_err(p.grad, sd[k].grad)
    flag = " <<<<" if e > 2e-3 else ""
    print("%-60s %.3e  |ref| %.3e%s" % (k, e, float(sd[k].grad.abs().max()), flag))
