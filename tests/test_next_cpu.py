"""CPU tier for the loop neighbours of SURVEY.md 8f (optimiser step, loss + accuracy epilogue,
input staging, LinearClassifier, NN retrieval):

  * the oracle's restatements against the fixtures recorded from the reference's own functions
    (oracle/make_golden_next.py -> tests/golden/next_*.pt);
  * the product's HOST logic (coclr_amd/{loss,staging,optim}.py, model/classifier.py,
    eval/retrieval.py) on the ATen test double of tests/fake_backend.py against the same fixtures.
The HIP kernels themselves are compared in tests/test_gpu_next.py."""
import os

import pytest
import torch
import torch.nn.functional as F

import fake_backend
from _cases import check_close, load_golden
from oracle import coclr_oracle as orc


@pytest.fixture
def fake(monkeypatch):
    fake_backend.install(monkeypatch)


# ---- oracle vs reference fixtures -----------------------------------------------------------

def test_oracle_loss_epilogue_matches_reference():
    for rec in load_golden("next_loss_epilogue")["cases"]:
        logits, mask, target = rec["logits"], rec["mask"], rec["target"]
        for name, fn in (("ce", lambda lg: F.cross_entropy(lg, target)),
                         ("multi", lambda lg: orc.multi_nce_loss(lg, mask)),
                         ("multi_drop", lambda lg: orc.masked_nce_loss_drop_self(lg, mask)),
                         ("uber", lambda lg: orc.ubernce_loss(lg, mask))):
            lg = logits.clone().requires_grad_(True)
            loss = fn(lg)
            loss.backward()
            check_close(loss, rec[name]["loss"], 1e-6, name + " loss")
            check_close(lg.grad, rec[name]["dlogits"], 1e-6, name + " dlogits")
        for got, ref in zip(orc.calc_topk_accuracy(logits, target, (1, 5)), rec["topk_self"]):
            assert float(got) == float(ref)
        for got, ref in zip(orc.calc_mask_accuracy(logits, mask, (1, 5)), rec["topk_mask"]):
            assert float(got) == float(ref)


def test_oracle_staging_matches_reference():
    g = load_golden("next_staging")
    out = orc.tr(g["u8"], g["num_seq"], g["seq_len"])
    assert torch.equal(out, g["out"])
    assert torch.equal(orc.tr(g["u8"].float() / 255, g["num_seq"], g["seq_len"]), g["out"])


def test_oracle_adam_matches_torch():
    g = load_golden("next_adam")
    ps = [p.double().clone() for p in g["p0"]]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    steps = [0] * len(ps)
    for step in range(3):
        orc.adam_step(ps, [x.double() for x in g["grads"][step]], ms, vs, steps, g["lr"],
                      g["betas"][0], g["betas"][1], g["eps"], g["wd"])
        for p, ref in zip(ps, g["after"][step]):
            # fp32 torch vs the float64 restatement: the update is ~lr, its fp32 rounding ~1e-7 of |p|
            check_close(p, ref, 2e-6, "adam step %d" % step)


def _classifier_block(g):
    return torch.randn(*g["block_shape"], generator=torch.Generator().manual_seed(g["block_seed"]))


def _classifier_state(g, module):
    cfg = g["cfg"]
    torch.manual_seed(21)
    clf = module.LinearClassifier(num_class=cfg["num_class"], network=cfg["network"], dropout=0.5,
                                  use_dropout=True, use_l2_norm=cfg["use_l2_norm"],
                                  use_final_bn=cfg["use_final_bn"])
    assert sorted(clf.state_dict()) == g["init_keys"]
    assert torch.equal(clf.state_dict()["final_fc.1.weight"], g["init_fc_weight"])
    tot = float(sum(v.double().sum() for v in clf.state_dict().values() if v.is_floating_point()))
    assert abs(tot - g["init_sum"]) < 1e-6 * max(1.0, abs(g["init_sum"]))
    return clf


def test_oracle_classifier_matches_reference():
    import model.classifier as product
    g = load_golden("next_classifier")
    clf = _classifier_state(g, product)        # same seed -> same init as the reference (asserted)
    sd = orc.training_state(clf.state_dict(), requires_grad_prefix="")
    with torch.no_grad():
        logit, feat = orc.linear_classifier_forward(sd, "s3d", _classifier_block(g), False, True, True)
    check_close(logit, g["logit_eval"], 2e-5, "eval logit")
    check_close(feat, g["feat_eval"], 2e-5, "eval feat")
    logit, feat = orc.linear_classifier_forward(sd, "s3d", _classifier_block(g), True, True, True)
    check_close(logit, g["logit_train"], 2e-5, "train logit")
    F.cross_entropy(logit, g["target"]).backward()
    for k, ref in g["grads"].items():
        check_close(sd[k].grad.reshape(-1)[:4096], ref, 1e-4, "grad " + k)
    check_close(sd["final_bn.running_var"], g["final_bn.running_var"], 1e-5, "running_var")


def test_oracle_retrieval_fixture_is_self_consistent():
    g = load_golden("next_retrieval")
    assert g["pinned"] is True            # eval/main_classifier.py:686-706 exec'd by line range (oracle/make_golden_next.py)
    acc, sim = orc.nn_retrieval(g["test_feature"], g["test_label"], g["train_feature"],
                                g["train_label"])
    assert acc == g["acc"] and torch.equal(sim, g["sim"])


# ---- product host logic on the test double ---------------------------------------------------

def test_loss_module_matches_reference(fake):
    from coclr_amd import loss as L
    for rec in load_golden("next_loss_epilogue")["cases"]:
        logits, mask, target = rec["logits"], rec["mask"], rec["target"]
        for name, fn in (("ce", lambda lg: L.CrossEntropyLoss()(lg, target)),
                         ("multi", lambda lg: L.multi_nce_loss(lg, mask)),
                         ("multi_drop", lambda lg: L.multi_nce_loss(lg, mask, drop_self=True)),
                         ("uber", lambda lg: L.ubernce_loss(lg, mask))):
            lg = logits.clone().requires_grad_(True)
            loss = fn(lg)
            assert loss.dim() == 0 and loss.requires_grad
            loss.backward()
            check_close(loss, rec[name]["loss"], 1e-6, name + " loss")
            check_close(lg.grad, rec[name]["dlogits"], 1e-5, name + " dlogits")
            # accuracy helpers reuse the statistics of the loss that has just run on `lg` ...
            calls = []
            orig = L.ops.nce_loss_fwd
            L.ops.nce_loss_fwd = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                if name == "ce":
                    t1, t5 = L.calc_topk_accuracy(lg, target, (1, 5))
                    assert not calls
                    assert [float(t1), float(t5)] == [float(v) for v in rec["topk_self"]]
                else:
                    m1, m5 = L.calc_mask_accuracy(lg, mask, (1, 5))
                    s1, s5 = L.calc_self_accuracy(lg, (1, 5))
                    assert not calls
                    assert [float(m1), float(m5)] == [float(v) for v in rec["topk_mask"]]
                    assert [float(s1), float(s5)] == [float(v) for v in rec["topk_self"]]
                    # ... and recompute when asked about something else
                    t1, t5 = L.calc_topk_accuracy(lg, target, (1, 5))
                    assert calls
                    assert [float(t1), float(t5)] == [float(v) for v in rec["topk_self"]]
            finally:
                L.ops.nce_loss_fwd = orig
        fresh = logits.clone()
        m1, m5 = L.calc_mask_accuracy(fresh, mask, (1, 5))      # no loss ran on `fresh`
        assert [float(m1), float(m5)] == [float(v) for v in rec["topk_mask"]]


def test_staging_module_matches_reference(fake):
    from coclr_amd import staging
    g = load_golden("next_staging")
    out = staging.tr(g["u8"], g["num_seq"], g["seq_len"])
    assert out.shape == g["out"].shape and torch.equal(out, g["out"])
    assert torch.equal(staging.tr(g["u8"].float() / 255, g["num_seq"], g["seq_len"]), g["out"])
    with pytest.raises(ValueError):
        staging.tr(g["u8"], 3, g["seq_len"])


def test_adam_patch_and_cpu_fallthrough():
    """The shim resolves torch.optim.Adam to the single-launch subclass; CPU parameters (not the
    product path) run torch's own implementation unchanged, state-dict format included."""
    import model.pretrain  # noqa: F401  (installs the subclass)
    from coclr_amd import optim as O
    assert torch.optim.Adam is O.ScopedAdam and issubclass(O.ScopedAdam, O.Adam) and \
        issubclass(O.Adam, O._TorchAdam)
    g = load_golden("next_adam")
    ps = [p.clone().requires_grad_(True) for p in g["p0"]]
    opt = torch.optim.Adam([{"params": p} for p in ps], lr=g["lr"], weight_decay=g["wd"])
    for step in range(3):
        for p, gr in zip(ps, g["grads"][step]):
            p.grad = gr.clone()
        opt.step()
        for p, ref in zip(ps, g["after"][step]):
            assert torch.equal(p.detach(), ref)
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and len(sd["param_groups"]) == 4
    # scope of the process-wide alias: parameters of nobody's registered module never reach the kernel
    assert opt._ours is False and opt._plan is None
    assert O.Adam([torch.nn.Parameter(torch.zeros(2))])._scoped is False
    m = model.pretrain.InfoNCE('s3d', 128, 32, 0.999, 0.07)
    own = torch.optim.Adam([{"params": p} for p in m.parameters()], lr=1e-3)
    assert O._owns_any(p for g_ in own.param_groups for p in g_["params"])


def test_classifier_host_logic_matches_reference(fake):
    import model.classifier as product
    g = load_golden("next_classifier")
    clf = _classifier_state(g, product)
    clf.eval()
    with torch.no_grad():
        logit, feat = clf(_classifier_block(g))
    check_close(logit, g["logit_eval"], 5e-4, "eval logit")
    check_close(feat, g["feat_eval"], 5e-4, "eval feat")
    clf.train()
    clf.final_fc[0].p = 0.0
    logit, feat = clf(_classifier_block(g))
    check_close(feat, g["feat_train"], 5e-4, "train feat")
    # final_bn normalises over THREE nearly identical rows (L2-normalised features of a randomly
    # initialised S3D differ by ~1e-3 between clips): the batch variance amplifies the double's
    # ~1e-5 backbone differences ~500x.  Well-conditioned BatchNorm1d checks live in the GPU tier.
    check_close(logit, g["logit_train"], 2e-2, "train logit")
    F.cross_entropy(logit, g["target"]).backward()
    named = dict(clf.named_parameters())
    for k in ("final_fc.1.weight", "final_fc.1.bias"):
        check_close(named[k].grad.reshape(-1)[:4096], g["grads"][k], 2e-2, "grad " + k)
    assert named["backbone.Conv_1a.conv1.weight"].grad is not None
    check_close(clf.final_bn.running_mean, g["final_bn.running_mean"], 1e-3, "running_mean")
    # dropout in training mode: identity in expectation, zeroes a share of the features
    clf.final_fc[0].p = 0.5
    torch.manual_seed(0)
    x = torch.ones(4, 1024)
    y = clf.final_fc[0](x)
    assert set(y.unique().tolist()) == {0.0, 2.0}


def test_retrieval_host_logic(fake):
    from coclr_amd.eval.retrieval import nn_retrieval
    g = load_golden("next_retrieval")
    acc, sim, topidx = nn_retrieval(g["test_feature"], g["test_label"], g["train_feature"],
                                    g["train_label"])
    check_close(sim, g["sim"], 1e-5, "sim")
    assert [round(float(a), 6) for a in acc] == [round(a, 6) for a in g["acc"]]
    assert topidx.shape == (g["test_feature"].shape[0], 50)
    with pytest.raises(ValueError):
        nn_retrieval(g["test_feature"], g["test_label"], g["train_feature"], g["train_label"], ks=(5, 1))


def test_ddp_default_for_this_model_only():
    """The shim flips DistributedDataParallel's `gradient_as_bucket_view` default to True for the
    InfoNCE / UberNCE / CoCLR modules only, and never overrides an explicit argument."""
    import os
    import torch.distributed as dist
    import model.pretrain as product
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29741")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        DDP = torch.nn.parallel.DistributedDataParallel
        torch.manual_seed(0)
        m = product.InfoNCE('s3d', 128, 32, 0.999, 0.07)
        assert DDP(m).gradient_as_bucket_view is True
        assert DDP(m, gradient_as_bucket_view=False).gradient_as_bucket_view is False
        assert DDP(torch.nn.Linear(3, 3)).gradient_as_bucket_view is False
    finally:
        if own:
            dist.destroy_process_group()


def test_gradients_are_written_into_ddp_buckets(fake, monkeypatch):
    """DDP glue (coclr_amd/parallel.py + engine.grad_out): once the communication hook has published
    the bucket views, a backward pass writes weight gradients straight into them -- `.grad` of the
    backbone parameters aliases the bucket storage and DDP's per-parameter copy (`aten::mul`) is not
    called for them -- and the parameters after four Adam steps are BIT-IDENTICAL to the run without
    the hook (COCLR_DDP_HOOK=0: DDP's own per-parameter path)."""
    import torch.distributed as dist
    import model.pretrain as product
    from coclr_amd import engine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29742")
    own = not dist.is_initialized()
    if own:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        def run(hook):
            monkeypatch.setenv("COCLR_DDP_HOOK", "1" if hook else "0")
            engine._GRAD_SLOTS.clear()
            torch.manual_seed(0)
            model = product.InfoNCE('s3d', 128, 32, 0.999, 0.07)
            ddp = torch.nn.parallel.DistributedDataParallel(model)
            opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3,
                                   weight_decay=1e-5)
            ddp.train()
            aliased, verified = [], []
            for step in range(4):
                g = torch.Generator().manual_seed(50 + step)
                block = torch.randn(4, 2, 3, 8, 32, 32, generator=g)
                torch.manual_seed(60 + step)
                out, tgt = ddp(block)
                loss = torch.nn.functional.cross_entropy(out, tgt)
                opt.zero_grad()
                loss.backward()
                slots = engine._GRAD_SLOTS
                n = 0
                for p in model.encoder_q[0].parameters():
                    s = slots.get(id(p))
                    if s is not None and p.grad is not None and \
                            p.grad.data_ptr() == s[1].data_ptr():
                        n += 1
                aliased.append(n)
                verified.append(sum(1 for p in model.encoder_q[0].parameters()
                                    if id(p) in engine._SLOTS_VERIFIED))
                opt.step()
            run.verified = verified
            return [p.detach().clone() for p in model.parameters()], aliased, (model, ddp)

        ref, _, _ = run(False)
        assert not engine._GRAD_SLOTS
        got, aliased, (model, ddp) = run(True)
        nparams = len(list(model.encoder_q[0].parameters()))
        # step 0 publishes the first buckets, DDP rebuilds them once after it, step 1 publishes the
        # rebuilt ones: from step 2 on every backbone gradient is produced in place
        assert aliased[-1] == nparams and aliased[-2] == nparams, (aliased, nparams)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
        # ... and a view counts as ACCEPTED by DDP (engine._SLOTS_VERIFIED: what an un-joined weight-gradient
        # stream requires, engine.Run.defer_side) only after a whole pass in which the bucket stayed at the
        # published address and every gradient in it was the alias: step 0 publishes, step 1 re-publishes the
        # rebuilt buckets (which un-verifies), step 2 is the first such pass
        assert run.verified[0] == 0 and run.verified[1] == 0 and run.verified[2] == nparams, run.verified
        p0 = next(model.encoder_q[0].parameters())
        engine.set_grad_slot(p0, engine._GRAD_SLOTS[id(p0)][1])          # a re-publication un-verifies
        assert id(p0) not in engine._SLOTS_VERIFIED
        # a caller that keeps `.grad` (zero_grad(set_to_none=False)) must NOT get the in-place path
        run_ = engine.Run(torch.device("cpu"), save=True)
        p = next(model.encoder_q[0].parameters())
        p.grad = torch.zeros_like(p)
        assert run_.grad_out(p).data_ptr() != engine._GRAD_SLOTS[id(p)][1].data_ptr()
        p.grad = None
        assert run_.grad_out(p).data_ptr() == engine._GRAD_SLOTS[id(p)][1].data_ptr()
        assert run_.grad_out(p).data_ptr() != engine._GRAD_SLOTS[id(p)][1].data_ptr()   # once per run
        # the slots belong to the wrapper: dropping it releases them (and the bucket storage they view)
        del ddp, run_
        import gc
        gc.collect()
        assert id(p) not in engine._GRAD_SLOTS
    finally:
        engine._GRAD_SLOTS.clear()
        if own:
            dist.destroy_process_group()
