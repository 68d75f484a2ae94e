"""Module-level parity on the MI355X: backbone building blocks and whole backbones run
through coclr_amd.engine, forward AND backward, against the CPU oracle on
well-conditioned random states (random BN affine, O(1) activations, >= 512 values per
BatchNorm channel) where fp32 gradients are reproducible."""
import pytest
import torch

from oracle import coclr_oracle as orc

import os

pytestmark = pytest.mark.gpu
VERBOSE = os.environ.get("COCLR_TEST_VERBOSE", "0") == "1"


def rel(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def randomise(module, seed):
    """Well-conditioned random state: conv weights ~ N(0, 1/sqrt(fan_in)), BN affine random."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.Conv3d):
            fan_in = m.weight[0].numel()
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (1.5 / fan_in ** 0.5)
        elif isinstance(m, torch.nn.BatchNorm3d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) + 0.5
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.3
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=g) + 0.5


def _oracle_run(module_sd, oracle_fn, x, dout, dtype, input_grad):
    sd = orc.training_state({"m." + k: v.cpu() for k, v in module_sd.items()}, "m.")
    if dtype == torch.float64:
        sd = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point()
                  else v) for k, v in sd.items()}
    xr = x.to(dtype).clone().requires_grad_(input_grad)
    ref = oracle_fn(sd, xr)
    ref.backward(dout.to(dtype))
    return sd, xr, ref


def check_module(module, oracle_fn, x, train=True, tol=1e-3, input_grad=True, factor=8.0,
                 floor=2e-3, per_tensor=False):
    """Forward+backward `module` on the GPU vs oracle_fn(sd, "m", x) on the CPU with identical
    state.  Forward / running statistics: fixed 1e-3.  Gradients: the product must be as
    close to a float64 evaluation as the fp32 CPU evaluation is, because ReLU/max-pool
    decisions at near-ties make deep-network gradients a discontinuous function of fp32
    round-off: the distribution of per-tensor errors (median, 90th percentile, mean) must stay
    within 3x the fp32 CPU evaluation's own (+ floor; `factor` for modules with < 20 tensors),
    and no tensor may be off by more than 0.3."""
    module.train(train)
    state = orc.training_state(module.state_dict(), requires_grad_prefix="\0")   # alias-preserving clone
    probe = oracle_fn(orc.training_state({"m." + k: v for k, v in state.items()}, "m."), x)
    torch.manual_seed(11)
    dout = torch.randn_like(probe)
    sd32, x32, ref32 = _oracle_run(state, oracle_fn, x, dout, torch.float32, input_grad)
    sd64, x64, ref64 = _oracle_run(state, oracle_fn, x, dout, torch.float64, input_grad)

    module = module.cuda()
    xg = x.cuda().requires_grad_(input_grad)
    out = module(xg)
    assert out.shape == ref32.shape
    out.backward(dout.cuda())
    assert rel(out, ref32) <= tol, "forward rel err %.3e" % rel(out, ref32)

    got_errs, ref_errs = [], []

    def bound(got, k32, k64, what):
        e_ref, e_got = rel(k32, k64), rel(got, k64)
        assert e_got <= 0.3, "%s: err vs fp64 %.3e (fp32 CPU: %.3e)" % (what, e_got, e_ref)
        got_errs.append(e_got)
        ref_errs.append(e_ref)
        if VERBOSE:
            print("   %-44s err vs fp64 %.3e   (fp32 CPU %.3e)" % (what, e_got, e_ref))
        return e_got, e_ref

    worst = ("", 0.0, 0.0)
    if input_grad:
        bound(xg.grad, x32.grad, x64.grad, "dx")
    for k, p in module.named_parameters():
        e_got, e_ref = bound(p.grad, sd32["m." + k].grad, sd64["m." + k].grad, "grad " + k)
        if e_got > worst[1]:
            worst = (k, e_got, e_ref)
    # Near-tie flips (a ReLU or max-pool decision at fp32 round-off distance from a tie) hit
    # single tensors hard: with 32..256 values per BatchNorm channel in the last stages one
    # flipped element moves that channel's gradients by percents.  Every fp32 evaluation flips
    # its OWN handful of elements (the CPU's mkldnn order, this library's tile order), so the
    # per-tensor ratio got/ref is meaningless; what must hold is that the product's errors are
    # DISTRIBUTED like the reference's: median, 90th percentile and mean within `factor_q` x
    # (+ floor), and no tensor beyond 0.3 (checked above).  The same block fed with identical
    # inputs agrees to 5e-4 everywhere (tools/debug_block.py).
    g, r = torch.tensor(got_errs), torch.tensor(ref_errs)
    n = len(got_errs)
    factor_q = 3.0 if n >= 20 else factor
    for q in ((0.5,) if per_tensor else (0.5, 0.9)):
        gq, rq = float(torch.quantile(g, q)), float(torch.quantile(r, q))
        assert gq <= factor_q * rq + floor, \
            "grad err quantile %.1f: %.3e vs fp32 CPU's own %.3e" % (q, gq, rq)
    assert float(g.mean()) <= factor_q * float(r.mean()) + floor, \
        "mean grad err vs fp64 %.3e, fp32 CPU's own %.3e" % (float(g.mean()), float(r.mean()))
    if per_tensor:
        # Well-conditioned cases (>= 512 values per BatchNorm channel): EVERY tensor is held, not
        # just the distribution -- no tensor beyond 2e-2 of the float64 gradient and at least 90 %
        # of them within max(3 x the fp32 CPU evaluation's own error, 5e-3), so that a wrong
        # weight-gradient path of a single layer (a 5 % error is 5e-2) cannot hide behind the
        # others.  Why not 1e-6 like the kernels themselves (profiles/r02_numerics_probe.txt: every
        # conv / BatchNorm kernel is within 7e-6 of float64 elementwise, like ATen's CPU kernels):
        # ONE ReLU decision taken differently -- an activation within fp32 round-off of zero, about
        # one per 1.5 M elements -- moves the bias gradient of its channel by one element of a
        # ~2048..8192-term sum, i.e. 2e-3..1e-2 of the tensor maximum, and every tensor upstream of
        # it with it (observed signature: exact zeros below the flipped layer, ~2e-3 above).
        ok = (g <= torch.maximum(3.0 * r, torch.tensor(5e-3))).float().mean()
        print("per-tensor gradient errors vs fp64: max %.2e median %.2e (fp32 CPU: max %.2e), "
              "%.0f %% within bound over %d tensors" % (float(g.max()), float(g.median()),
                                                      float(r.max()), 100 * float(ok), n))
        assert float(g.max()) <= 2e-2, "worst tensor %s: %.3e (fp32 CPU: %.3e)" % worst
        assert float(ok) >= 0.90, "only %.0f %% of the tensors within max(3x CPU, 5e-3)" % (100 * float(ok))
    if train:
        for k, v in module.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                assert rel(v, sd32["m." + k]) <= tol, k
            if k.endswith("num_batches_tracked"):
                assert int(v) == int(sd32["m." + k]), k
    return rel(out, ref32), worst


def test_basic_and_separable_conv_units():
    from backbone.s3dg import BasicConv3d, STConv3d
    torch.manual_seed(0)
    m = BasicConv3d(24, 40, kernel_size=1, stride=1)
    randomise(m, 1)
    x = torch.randn(4, 24, 4, 8, 8)
    check_module(m, lambda sd, xx: orc.basic_conv3d(sd, "m", xx, True), x, per_tensor=True)
    m = STConv3d(3, 64, kernel_size=7, stride=2, padding=3)
    randomise(m, 2)
    x = torch.randn(2, 3, 8, 32, 32)
    check_module(m, lambda sd, xx: orc.st_conv3d(sd, "m", xx, True, 2, 3), x, per_tensor=True)
    m = STConv3d(32, 48, kernel_size=3, stride=1, padding=1)
    randomise(m, 3)
    x = torch.randn(3, 32, 4, 8, 8)
    check_module(m, lambda sd, xx: orc.st_conv3d(sd, "m", xx, True, 1, 1), x, per_tensor=True)


@pytest.mark.parametrize("gating", [False, True])
def test_sep_inception_block(gating):
    from backbone.s3dg import SepInception
    torch.manual_seed(0)
    m = SepInception(in_planes=48, out_planes=[16, 24, 32, 8, 16, 24], gating=gating)
    randomise(m, 4)
    x = torch.relu(torch.randn(4, 48, 4, 8, 8))      # post-ReLU-like input (ties for the pool)
    # oracle keys: "<pre>.branch0.0..." -> strip the leading dot by using pre="m" then renaming
    check_module(m, lambda sd, xx: orc.sep_inception(sd, "m", xx, True, gating), x, per_tensor=True)


def test_s3d_stages_per_tensor_gradients():
    """Stages 2 and 3 of S3D at their real widths (Conv_2b/2c; MaxPool_3a + Mixed_3b + Mixed_3c,
    backbone/s3dg.py:151-164) with >= 2048 values per BatchNorm channel: every one of their 54
    parameter tensors individually within bound -- direct, Winograd (spatial and temporal), fused
    pointwise heads, pooling, gradient accumulation over fan-out, all wgrad kernels."""
    import torch.nn.functional as F
    from backbone.s3dg import S3D
    torch.manual_seed(0)
    net = S3D()
    randomise(net, 8)
    x = torch.relu(torch.randn(4, 64, 8, 32, 32))
    check_module(net.block2, lambda sd, xx: orc.st_conv3d(
        sd, "m.2", orc.basic_conv3d(sd, "m.1", F.max_pool3d(xx, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                    True), True, 1, 1), x, per_tensor=True)
    x = torch.relu(torch.randn(4, 192, 8, 32, 32))
    check_module(net.block3, lambda sd, xx: orc.sep_inception(
        sd, "m.2", orc.sep_inception(sd, "m.1", F.max_pool3d(xx, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                     True, False), True, False), x, per_tensor=True)


def test_eval_mode_units_fold_bn_and_support_grad():
    from backbone.s3dg import STConv3d
    torch.manual_seed(0)
    m = STConv3d(16, 32, kernel_size=3, stride=1, padding=1)
    randomise(m, 5)
    x = torch.randn(2, 16, 4, 8, 8)
    # eval-mode BN with gradients (frozen statistics fine-tuning)
    check_module(m, lambda sd, xx: orc.st_conv3d(sd, "m", xx, False, 1, 1), x, train=False)
    # eval-mode, no grad: single-pass conv+affine+ReLU epilogue
    m = m.cuda().eval()
    sd = {"m." + k: v.cpu() for k, v in m.state_dict().items()}
    with torch.no_grad():
        out = m(x.cuda())
        ref = orc.st_conv3d(sd, "m", x, False, 1, 1)
    assert rel(out, ref) <= 1e-3


def test_s3d_backbone_forward_backward():
    from backbone.s3dg import S3D
    torch.manual_seed(0)
    m = S3D()
    randomise(m, 6)
    x = torch.randn(4, 3, 16, 64, 64)
    e, worst = check_module(m, lambda sd, xx: orc.s3d_forward(sd, "m.", xx, True), x,
                            input_grad=False)
    print("S3D fwd rel err %.2e, worst param grad %s %.2e (fp32 CPU: %.2e)" % ((e,) + worst))


def test_r50_backbone_forward_backward():
    from backbone.resnet_2d3d import r2d3d50
    torch.manual_seed(0)
    m = r2d3d50()
    randomise(m, 7)
    x = torch.randn(2, 3, 8, 64, 64)
    e, worst = check_module(m, lambda sd, xx: orc.r2d3d50_forward(sd, "m.", xx, True), x,
                            input_grad=False)
    print("r50 fwd rel err %.2e, worst param grad %s %.2e (fp32 CPU: %.2e)" % ((e,) + worst))


def test_backbone_input_gradient_and_no_grad_paths():
    from backbone.s3dg import S3D
    torch.manual_seed(0)
    m = S3D().cuda().train()
    x = torch.randn(2, 3, 16, 64, 64, device="cuda", requires_grad=True)
    y = m(x)
    y.sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad).all()
    with torch.no_grad():
        y2 = m(x)
    assert not y2.requires_grad and y2.shape == y.shape
    # strided clip view (block[:, i]) is consumed without a copy and gives the same result
    blk = torch.randn(2, 2, 3, 16, 64, 64, device="cuda")
    m.eval()
    with torch.no_grad():
        a = m(blk[:, 1])
        b = m(blk[:, 1].contiguous())
    assert torch.equal(a, b)
