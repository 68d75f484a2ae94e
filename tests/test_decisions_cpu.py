"""CPU tier for the decision-conditioned gradient comparison (tests/_decisions.py).

(1) The premise, on the oracle alone: the fp32-vs-fp64 gradient error of a de-saturated S3D InfoNCE
    step is made of discrete ReLU / max-pool decision flips; with the fp32 run's decisions handed to
    the float64 run, what is left is round-off (1e-4), orders of magnitude below the raw error.
(2) The plumbing: the product (on the ATen test double) reports its decisions through the engine's
    probe under the reference's unit names, and the oracle forced onto them reproduces the product's
    gradients tensor by tensor."""
import torch

import fake_backend
from _cases import build_model, case_inputs
from _decisions import l2_table, oracle_grads, product_grads, record_product

CFG = dict(kind="infonce", network="s3d", B=4, K=32, dim=128, m=0.999, T=0.07, clip=(3, 16, 64, 64),
           model_seed=0, input_seed=21, perm_seed=120, condition=dict(seed=9))


def _setup():
    import model.pretrain as product
    model = build_model(CFG, product)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    blocks, extra = case_inputs(CFG, 0)
    torch.manual_seed(CFG["perm_seed"])
    perm = torch.randperm(CFG["B"])
    return model, sd0, blocks, extra, perm


def _median(vals):
    vals = sorted(vals)
    return vals[len(vals) // 2]


def test_fp32_gradient_error_is_decision_flips():
    _, sd0, blocks, extra, perm = _setup()
    g32, _, _, d32 = oracle_grads(sd0, CFG, blocks, extra, perm, torch.float32)
    g64, _, _, d64 = oracle_grads(sd0, CFG, blocks, extra, perm, torch.float64)
    g64f, _, _, _ = oracle_grads(sd0, CFG, blocks, extra, perm, torch.float64, decisions=d32)
    assert len(d32.relu) == 78 and len(d32.pools) == 13 and "head" in d32.relu
    assert set(d32.relu) == set(d64.relu)
    flips = d32.flips(d64)
    nflips = sum(f[1] for f in flips)
    raw = _median(v[0] for v in l2_table(g32, g32, g64).values())
    smooth = l2_table(g32, g32, g64f)
    med_smooth, max_smooth = _median(v[0] for v in smooth.values()), max(v[0] for v in smooth.values())
    print("flips %d of %d decisions; median L2 error raw %.2e, decision-conditioned %.2e (max %.2e)"
          % (nflips, d32.count(), raw, med_smooth, max_smooth))
    assert max_smooth < 2e-3, "round-off alone should leave every tensor within 2e-3"
    if nflips:
        assert med_smooth < raw, "conditioning on the decisions must not increase the error"


def test_product_decisions_reproduce_on_the_oracle(monkeypatch):
    fake_backend.install(monkeypatch)
    model, sd0, blocks, extra, perm = _setup()
    model.train()

    def step():
        torch.manual_seed(CFG["perm_seed"])
        out, tgt = model(blocks[0])
        loss = torch.nn.functional.cross_entropy(out, tgt)
        loss.backward()
        return out

    out, dec = record_product(model, step)
    assert len(dec.relu) == 78 and len(dec.pools) == 13
    ref, _, ref_logits, _ = oracle_grads(sd0, CFG, blocks, extra, perm, torch.float32, decisions=dec)
    got = product_grads(model)
    assert set(got) == set(ref) and len(got) >= 235
    assert float((out.detach() - ref_logits).abs().max() / ref_logits.abs().max()) < 1e-3
    table = l2_table(got, got, ref)
    worst = max(table.items(), key=lambda kv: kv[1][0])
    print("worst tensor %s: L2 error %.2e; median %.2e" % (worst[0], worst[1][0],
                                                          _median(v[0] for v in table.values())))
    assert worst[1][0] < 2e-3, worst
