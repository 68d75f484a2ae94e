"""Kernel-level parity: every HIP kernel (called through the C ABI via
coclr_amd.ops) against the same op in plain fp32 PyTorch on the CPU.

Tolerances: fp32 accumulation order differs from ATen's, so values are compared
with max|delta| <= 2e-4 * max|ref| (well inside BASELINE.json's 1e-3); integer /
index / mask outputs must match exactly.
"""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-4


def close(got, ref, rtol=RTOL, what=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rtol * scale, "%s: max err %.3e vs scale %.3e (rel %.2e)" % (what, err, scale,
                                                                            err / scale)


def dev(t):
    return t.cuda()


def run_conv(x, w, stride, padding, accumulate_into=None, n_index=None, want_stats=True):
    from coclr_amd import ops, engine
    N = x.shape[0] if n_index is None else n_index.shape[0]
    g = ops.ConvGeom(N, x.shape[1], w.shape[0], x.shape[2:], w.shape[2:], stride, padding)
    run = engine.Run(torch.device("cuda"), save=False)
    wd = dev(w)
    packed = run.pack(wd, False)
    y = torch.zeros(N, w.shape[0], *g.odim, device="cuda") if accumulate_into is None \
        else dev(accumulate_into)
    stats = torch.empty(2 * w.shape[0] * g.ntiles(), device="cuda") if want_stats else None
    ops.conv_fwd(g, dev(x), packed, y, stats=stats,
                 n_index=dev(n_index) if n_index is not None else None,
                 accumulate=accumulate_into is not None)
    torch.cuda.synchronize()
    return g, y, stats


CONV_CASES = [
    # (N, Cin, Cout, (T,H,W), k, s, p)
    (2, 3, 64, (8, 32, 32), (1, 7, 7), (1, 2, 2), (0, 3, 3)),      # stem spatial
    (2, 64, 64, (8, 16, 16), (7, 1, 1), (2, 1, 1), (3, 0, 0)),     # stem temporal
    (2, 64, 64, (4, 16, 16), (1, 1, 1), (1, 1, 1), (0, 0, 0)),     # Conv_2b
    (2, 64, 192, (4, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)),    # Conv_2c spatial
    (2, 192, 192, (8, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)),     # Conv_2c temporal
    (3, 96, 208, (4, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # ragged Cout
    (3, 208, 208, (4, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (5, 480, 16, (2, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)),      # tiny Cout, odd batch
    (2, 16, 48, (2, 4, 4), (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # block5-like extent
    (2, 48, 48, (2, 4, 4), (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (2, 24, 64, (3, 7, 7), (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # non power-of-two extent
    (2, 64, 64, (3, 7, 7), (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (2, 832, 384, (2, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0)),     # Mixed_5c.branch0
    (2, 128, 128, (4, 14, 14), (1, 3, 3), (1, 2, 2), (0, 1, 1)),   # r50 strided conv2
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "%d_%d_%d_%s_%s" % (c[0], c[1], c[2], c[3], c[4]))
def test_conv_forward_and_stats(case):
    N, Cin, Cout, dims, k, s, p = case
    torch.manual_seed(0)
    x = torch.randn(N, Cin, *dims)
    w = torch.randn(Cout, Cin, *k) * 0.05
    ref = F.conv3d(x, w, None, s, p)
    g, y, stats = run_conv(x, w, s, p)
    close(y, ref, what="conv y")
    st = stats.view(2, Cout, -1).double().sum(-1).cpu()
    close(st[0], ref.double().sum((0, 2, 3, 4)), rtol=1e-3, what="stats sum")
    close(st[1], (ref.double() ** 2).sum((0, 2, 3, 4)), what="stats sumsq")


@pytest.mark.parametrize("case", CONV_CASES[:13], ids=lambda c: "%d_%d_%d_%s_%s" % (c[0], c[1], c[2], c[3], c[4]))
def test_conv_dgrad_wgrad(case):
    from coclr_amd import ops, engine
    N, Cin, Cout, dims, k, s, p = case
    torch.manual_seed(1)
    x = torch.randn(N, Cin, *dims, requires_grad=True)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, s, p)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p)
    run = engine.Run(torch.device("cuda"), save=False)
    wd, dyd, xd = dev(w.detach()), dev(dy), dev(x.detach())
    # data gradient, written then accumulated
    dx = torch.empty(N, Cin, *dims, device="cuda")
    ops.conv_fwd(g.dgrad(), dyd, run.pack(wd, True), dx)
    close(dx, x.grad, what="dgrad")
    ops.conv_fwd(g.dgrad(), dyd, run.pack(wd, True), dx, accumulate=True)
    close(dx, 2 * x.grad, what="dgrad accumulate")
    # weight gradient
    dw = torch.empty_like(wd)
    ws = torch.empty(g.wgrad_workspace(), device="cuda")
    kk = k[0] * k[1] * k[2]
    ops.conv_wgrad(g, xd, dyd, dw, ws, Cin * kk, kk, 0)
    close(dw, w.grad, what="wgrad")


@pytest.mark.parametrize("relu,training", [(True, True), (False, True), (True, False)],
                         ids=["bn_relu_train", "bn_train", "bn_relu_frozen"])
@pytest.mark.parametrize("N,dims", [(2, (8, 32, 32)), (3, (5, 30, 26)), (3, (11, 70, 66)), (5, (8, 64, 64))],
                         ids=["even", "ragged", "ragged_streaming", "streaming"])
def test_stem_weight_gradient_applies_batchnorm_backward(N, dims, relu, training):
    """coclr_conv3d_wgrad_bn (the (1,7,7) stem, backbone/s3dg.py:145): the weight gradient formed straight from
    d(activation) and the conv output, BatchNorm's backward applied between LDS and the matrix pipe -- against
    autograd on the CPU, and BIT-IDENTICAL to the two-pass form (bn_act_backward then conv_wgrad) wherever that
    form runs the same reduction (channels of more than ops.SMALL_CHANNEL values: the streaming kernels)."""
    from coclr_amd import ops
    torch.manual_seed(11)
    Cin, Cout, k, s, p = 3, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)
    x = torch.randn(N, Cin, *dims)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    gamma = (torch.rand(Cout) + 0.5).requires_grad_(True)
    beta = (torch.randn(Cout) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(Cout) * 0.1, torch.rand(Cout) + 0.5
    y = F.conv3d(x, w, None, s, p)
    z = F.batch_norm(y, rm.clone(), rv.clone(), gamma, beta, training, 0.1, 1e-5)
    if relu:
        z = torch.relu(z)
    dz = torch.randn_like(z)
    z.backward(dz)

    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p)
    assert g.wgrad_bn_ok()
    yd, xd = dev(y.detach()), dev(x)
    if training:
        mean = y.detach().double().mean((0, 2, 3, 4))
        var = y.detach().double().var((0, 2, 3, 4), unbiased=False)
    else:
        mean, var = rm.double(), rv.double()
    invstd = (var + 1e-5).rsqrt()
    scale = gamma.detach().double() * invstd
    shift = beta.detach().double() - mean * scale
    small = dev(torch.stack([mean, invstd, scale, shift]).float())
    # d(activation) lives in a channel slice of a wider tensor (sample stride != C * S)
    dzw = torch.zeros(N, Cout + 5, *g.odim, device="cuda")
    dzd = dzw[:, 2:2 + Cout]
    dzd.copy_(dev(dz))
    nws = ops.bn_backward_workspace(N, Cout)
    kk = k[0] * k[1] * k[2]
    # two passes: dy written, then read
    dy = torch.empty_like(yd)
    dgb = torch.empty(2, Cout, device="cuda")
    ops.bn_act_backward(dzd, yd, None, small[2], small[3], small[0], small[1],
                        torch.full((nws,), float("nan"), dtype=torch.float64, device="cuda"), dy, None, dgb[0],
                        dgb[1], relu, training)
    dw2 = torch.empty(Cout, Cin, *k, device="cuda")
    ops.conv_wgrad(g, xd, dy, dw2, torch.empty(g.wgrad_workspace(), device="cuda"), Cin * kk, kk, 0)
    # one: coefficients, then the weight gradient applies them
    coef = torch.full((5, Cout), float("nan"), device="cuda")
    dgb1 = torch.empty(2, Cout, device="cuda")
    ops.bn_act_backward_coeffs(dzd, yd, small[2], small[3], small[0], small[1],
                               torch.full((nws,), float("nan"), dtype=torch.float64, device="cuda"), coef,
                               dgb1[0], dgb1[1], relu, training)
    dw1 = torch.full((Cout, Cin, *k), float("nan"), device="cuda")
    ops.conv_wgrad_bn(g, xd, dzd, yd, coef, relu, dw1, torch.empty(g.wgrad_workspace(), device="cuda"),
                      Cin * kk, kk)
    torch.cuda.synchronize()
    if N * g.odim[0] * g.odim[1] * g.odim[2] > ops.SMALL_CHANNEL:
        assert torch.equal(dgb1, dgb)
        assert torch.equal(dw1, dw2)
    else:
        close(dgb1, dgb, rtol=1e-5, what="sums, one-launch form")
        close(dw1, dw2, rtol=1e-5, what="wgrad, one-launch form")
    close(dw1, w.grad, rtol=5e-4, what="stem wgrad through BatchNorm")
    close(dgb1[0], gamma.grad, rtol=5e-4, what="dgamma")
    close(dgb1[1], beta.grad, rtol=5e-4, what="dbeta")
    # not a stem: no such kernel, and the call says so
    g2 = ops.ConvGeom(2, 64, 64, (4, 16, 16), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    assert not g2.wgrad_bn_ok()


@pytest.mark.parametrize("wino", [True, False], ids=["winograd_phases", "direct_phases"])
@pytest.mark.parametrize("T", [8, 9, 32, 34, 37])
def test_conv_dgrad_phase_decomposition(T, wino, monkeypatch):
    """Strided temporal stem conv (backbone/s3dg.py:41): its data gradient as two dense
    stride-1 correlations writing the even / odd frames of dX, plain and accumulated.  From 16 frames per phase
    up the 3-tap phase runs through F(4,3) and the 4-tap one through F(2,4) (algo = 2 with a destination lattice
    along T; COCLR_WINO_PHASES=0 keeps the direct kernels)."""
    from coclr_amd import ops, engine
    monkeypatch.setattr(ops, "WINOGRAD_PHASES", wino)
    N, Cin, Cout, dims, k, s, p = 2, 64, 72, (T, 12, 12), (7, 1, 1), (2, 1, 1), (3, 0, 0)
    torch.manual_seed(5)
    x = torch.randn(N, Cin, *dims, requires_grad=True)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, s, p)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p)
    phases = g.dgrad_phases()
    assert phases is not None and len(phases) == 2
    assert sorted(nk for _, _, nk, _ in phases) == [3, 4]
    for pg, _, nk, _ in phases:
        # a phase is a 'same' convolution over dY's frames only when T is even; odd T keeps the direct kernels
        assert pg.algo == (2 if (wino and pg.odim[0] >= 16 and pg.odim[0] == pg.idim[0]) else 0), (nk, pg)
    run = engine.Run(torch.device("cuda"), save=False)
    wd, dyd = dev(w.detach()), dev(dy)
    dx = torch.full((N, Cin, *dims), float("nan"), device="cuda")
    for pg, k0, nk, step in phases:
        ops.conv_fwd(pg, dyd, run.pack(wd, True, taps=nk, tap_base=k0, tap_step=step, algo=pg.algo), dx)
    close(dx, x.grad, what="phase dgrad")
    for pg, k0, nk, step in phases:
        ops.conv_fwd(pg, dyd, run.pack(wd, True, taps=nk, tap_base=k0, tap_step=step, algo=pg.algo), dx,
                     accumulate=True)
    close(dx, 2 * x.grad, what="phase dgrad accumulate")
    # geometries outside the supported form fall back to the dilated formulation
    assert ops.ConvGeom(2, 8, 8, (4, 14, 14), (1, 3, 3), (1, 2, 2), (0, 1, 1)).dgrad_phases() is None
    assert ops.ConvGeom(2, 8, 8, (4, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)).dgrad_phases() is None


def test_conv_channel_slices_gather_and_epilogue():
    """x read from a wider buffer through n_index; y accumulated; fused affine+ReLU."""
    from coclr_amd import ops, engine
    torch.manual_seed(2)
    big = torch.randn(6, 40, 4, 8, 8)
    x = big[:, 8:24]
    w = torch.randn(32, 16, 1, 3, 3) * 0.1
    idx = torch.tensor([4, 0, 5, 2])
    ref = F.conv3d(x[idx], w, None, 1, (0, 1, 1))
    bigd = dev(big)
    g, y, _ = run_conv(bigd[:, 8:24], w, (1, 1, 1), (0, 1, 1), n_index=idx)
    close(y, ref, what="gathered slice conv")
    base = torch.randn_like(ref)
    g, y2, st = run_conv(bigd[:, 8:24], w, (1, 1, 1), (0, 1, 1), accumulate_into=base, n_index=idx)
    close(y2, ref + base, what="accumulate")
    close(st.view(2, 32, -1).sum(-1)[0], (ref + base).sum((0, 2, 3, 4)), rtol=1e-3, what="acc stats")
    # epilogue: scale/shift/relu into a channel slice of a wider output
    run = engine.Run(torch.device("cuda"), save=False)
    out = torch.zeros(4, 50, 4, 8, 8, device="cuda")
    sc, sh = torch.rand(32) + 0.5, torch.randn(32)
    gg = ops.ConvGeom(4, 16, 32, (4, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1))
    ops.conv_fwd(gg, bigd[:, 8:24], run.pack(dev(w), False), out[:, 10:42], ep_scale=dev(sc),
                 ep_shift=dev(sh), n_index=dev(idx), relu=True)
    exp = torch.relu(ref * sc[None, :, None, None, None] + sh[None, :, None, None, None])
    close(out[:, 10:42], exp, what="epilogue")
    assert out[:, :10].abs().max().item() == 0 and out[:, 42:].abs().max().item() == 0


def test_sliced_stem_5x7x7():
    """(5,7,7)/2 conv as five accumulated (1,7,7) launches + sliced wgrad (r50 stem)."""
    from coclr_amd import ops, engine
    torch.manual_seed(3)
    x = torch.randn(2, 3, 8, 32, 32)
    w = (torch.randn(64, 3, 5, 7, 7) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, (2, 2, 2), (2, 3, 3))
    dyr = torch.randn_like(ref)
    ref.backward(dyr)
    run = engine.Run(torch.device("cuda"), save=False)
    xd, wd, dyd = dev(x), dev(w.detach()), dev(dyr)
    odim = tuple(ref.shape[2:])
    y = torch.empty(2, 64, *odim, device="cuda")
    dw = torch.zeros_like(wd)
    for t in range(5):
        g = ops.ConvGeom(2, 3, 64, x.shape[2:], (1, 7, 7), (2, 2, 2), (2 - t, 3, 3), odim=odim)
        ops.conv_fwd(g, xd, run.pack(wd, False, t), y, accumulate=t > 0)
        ws = torch.empty(g.wgrad_workspace(), device="cuda")
        ops.conv_wgrad(g, xd, dyd, dw, ws, 3 * 245, 245, t * 49)
    close(y, ref.detach(), what="sliced stem fwd")
    close(dw, w.grad, what="sliced stem wgrad")


@pytest.mark.parametrize("shape,relu,res", [((4, 24, 4, 8, 8), True, False),
                                            ((3, 10, 3, 5, 7), True, True),
                                            ((2, 64, 2, 4, 4), False, False),
                                            ((4, 8, 8, 32, 32), True, False),
                                            ((3, 6, 7, 30, 31), True, True),
                                            ((3, 6, 7, 30, 31), True, False),
                                            # N*S > 32768 per channel: the streaming two-pass kernels
                                            ((2, 4, 16, 32, 40), True, False),
                                            ((5, 3, 9, 31, 33), False, False)])
def test_batchnorm_train_fwd_bwd(shape, relu, res):
    from coclr_amd import ops
    torch.manual_seed(4)
    N, Cc = shape[0], shape[1]
    y = (torch.randn(*shape) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(Cc) + 0.5).requires_grad_(True)
    beta = torch.randn(Cc).requires_grad_(True)
    rm, rv = torch.randn(Cc), torch.rand(Cc) + 0.5
    rm0, rv0 = rm.clone(), rv.clone()
    resid = torch.randn(*shape).requires_grad_(True) if res else None
    z = F.batch_norm(y, rm, rv, gamma, beta, True, 0.1, 1e-5)
    if res:
        z = z + resid
    if relu:
        z = torch.relu(z)
    dz = torch.randn_like(z)
    z.backward(dz)

    yd = dev(y.detach())
    S = shape[2] * shape[3] * shape[4]
    # statistics as the conv epilogue would deliver them: 3 partial tiles per channel
    yy = y.detach().double().transpose(0, 1).reshape(Cc, -1)
    chunks = torch.chunk(yy, 3, dim=1)
    stats = torch.stack([torch.stack([c.sum(1) for c in chunks], 1),
                         torch.stack([(c ** 2).sum(1) for c in chunks], 1)]).float()
    small = torch.empty(4, Cc, device="cuda")
    rmd, rvd = dev(rm0), dev(rv0)
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    ops.bn_finalize(dev(stats).contiguous(), Cc, 3, N * S, dev(gamma.detach()), dev(beta.detach()),
                    rmd, rvd, nbt, 0.1, 1e-5, small[0], small[1], small[2], small[3])
    close(rmd, rm, what="running_mean")
    close(rvd, rv, what="running_var")
    assert int(nbt) == 1
    wide = torch.zeros(N, Cc + 6, *shape[2:], device="cuda")
    zd = wide[:, 3:3 + Cc]
    rd = dev(resid.detach()) if res else None
    ops.bn_act_apply(yd, small[2], small[3], rd, zd, relu)
    close(zd, z, what="bn apply")
    # backward; dz lives in a channel slice too
    dzw = torch.zeros(N, Cc + 6, *shape[2:], device="cuda")
    dzw[:, 3:3 + Cc] = dev(dz)
    dy = torch.empty_like(yd)
    dgb = torch.empty(2, Cc, device="cuda")
    dres = torch.full(shape, 1.0, device="cuda") if res else None
    ops.bn_act_backward(dzw[:, 3:3 + Cc], yd, zd if res else None, small[2], small[3], small[0],
                        small[1], torch.full((ops.bn_backward_workspace(N, Cc),), float("nan"),
                                             dtype=torch.float64, device="cuda"),
                        dy, dres, dgb[0], dgb[1], relu, True, dres_accumulate=res)
    close(dy, y.grad, rtol=5e-4, what="bn dy")
    close(dgb[0], gamma.grad, rtol=5e-4, what="dgamma")
    close(dgb[1], beta.grad, rtol=5e-4, what="dbeta")
    if res:
        close(dres, resid.grad + 1.0, what="dres accumulate")
    else:
        # statistics + apply as ONE call (a single launch when a channel holds <= 32768 values,
        # two launches above): identical result, running statistics advanced once
        rm2, rv2 = dev(rm0), dev(rv0)
        nbt2 = torch.zeros((), dtype=torch.long, device="cuda")
        small2 = torch.empty(4, Cc, device="cuda")
        wide2 = torch.zeros(N, Cc + 6, *shape[2:], device="cuda")
        ops.bn_finalize_apply(dev(stats).contiguous(), Cc, 3, N * S, dev(gamma.detach()),
                              dev(beta.detach()), rm2, rv2, nbt2, 0.1, 1e-5, small2[0], small2[1],
                              small2[2], small2[3], yd, wide2[:, 3:3 + Cc], relu)
        assert torch.equal(small2, small) and torch.equal(rm2, rmd) and torch.equal(rv2, rvd)
        assert int(nbt2) == 1 and torch.equal(wide2, wide)


def test_batchnorm_eval_affine():
    from coclr_amd import ops
    torch.manual_seed(5)
    Cc = 20
    y = torch.randn(2, Cc, 2, 4, 4)
    gamma, beta, rm, rv = torch.rand(Cc) + 0.5, torch.randn(Cc), torch.randn(Cc), torch.rand(Cc) + 0.5
    ref = torch.relu(F.batch_norm(y, rm, rv, gamma, beta, False, 0.1, 1e-5))
    small = torch.empty(4, Cc, device="cuda")
    ops.bn_eval_affine(dev(gamma), dev(beta), dev(rm), dev(rv), 1e-5, Cc, small[0], small[1],
                       small[2], small[3])
    z = torch.empty(2, Cc, 2, 4, 4, device="cuda")
    ops.bn_act_apply(dev(y), small[2], small[3], None, z, True)
    close(z, ref, what="eval bn")


POOLS = [((1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 5, 4, 16, 16)),
         ((3, 3, 3), (2, 2, 2), (1, 1, 1), (2, 6, 8, 8, 8)),
         ((2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 7, 4, 4, 4)),
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (3, 4, 4, 6, 5)),
         ((1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 3, 2, 7, 7)),
         # inception pool branch on power-of-two planes: the separable kernel (thread = (volume, h, w))
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 5, 16, 16, 16)),
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (3, 7, 8, 8, 8)),
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 37, 4, 4, 4)),
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 3, 1, 4, 4)),
         ((3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 3, 5, 8, 4))]


@pytest.mark.parametrize("k,s,p,shape", POOLS)
def test_maxpool_fwd_bwd(k, s, p, shape):
    from coclr_amd import ops
    torch.manual_seed(6)
    # post-ReLU-like input: many exact ties at 0 exercise first-max-wins
    x = torch.relu(torch.randn(*shape)).requires_grad_(True)
    ref, ridx = F.max_pool3d(x, k, s, p, return_indices=True)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    g = ops.PoolGeom(shape[0], shape[1], shape[2:], k, s, p)
    y = torch.empty(shape[0], shape[1], *g.odim, device="cuda")
    idx = torch.empty(shape[0], shape[1], *g.odim, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd(g, dev(x.detach()), y, idx)
    assert torch.equal(y.cpu(), ref.detach()), "maxpool values must be exact"
    assert torch.equal(idx.cpu().long(), ridx), "argmax must match ATen's first-max-wins"
    dx = torch.full(shape, 0.5, device="cuda")
    ops.maxpool_bwd(g, dev(dy), idx, dx, accumulate=True)
    close(dx, x.grad + 0.5, what="maxpool bwd accumulate")
    ops.maxpool_bwd(g, dev(dy), idx, dx, accumulate=False)
    close(dx, x.grad, what="maxpool bwd")
    # without indices (no-grad encoders), into a channel slice of a wider buffer
    wide = torch.zeros(shape[0], shape[1] + 3, *g.odim, device="cuda")
    ops.maxpool_fwd(g, dev(x.detach()), wide[:, 2:2 + shape[1]])
    assert torch.equal(wide[:, 2:2 + shape[1]].cpu(), ref.detach())
    assert float(wide[:, :2].abs().max()) == 0 and float(wide[:, 2 + shape[1]:].abs().max()) == 0
    # values with -inf / NaN: ATen's scan semantics (NaN wins, all -inf keeps the first element)
    xs = torch.randn(*shape)
    xs.view(-1)[::7] = -float("inf")
    xs.view(-1)[5::131] = float("nan")
    if xs[0, 0].numel() >= 8:
        xs[0, 0].fill_(-float("inf"))
    ref2, ridx2 = F.max_pool3d(xs, k, s, p, return_indices=True)
    ops.maxpool_fwd(g, dev(xs), y, idx)
    assert torch.equal(torch.nan_to_num(y.cpu(), nan=123.0), torch.nan_to_num(ref2, nan=123.0))
    assert torch.equal(idx.cpu().long(), ridx2)


@pytest.mark.parametrize("k,s,p,shape", POOLS[:3] + POOLS[5:7])
def test_maxpool_with_fused_batchnorm_relu_input(k, s, p, shape):
    """max_pool(relu(y*scale + shift)) with the affine + ReLU applied while the pool reads y (the
    BatchNorm unit in front of MaxPool_2a / MaxPool_3a, backbone/s3dg.py:151,162): identical to the
    two-pass form, negative scales included."""
    from coclr_amd import ops
    torch.manual_seed(7)
    yraw = torch.randn(*shape)
    scale = torch.randn(shape[1])                    # both signs
    shift = torch.randn(shape[1]) * 0.3
    z = torch.relu(yraw * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1))
    zd = torch.empty(shape, device="cuda")
    ops.bn_act_apply(dev(yraw), dev(scale), dev(shift), None, zd, True)
    g = ops.PoolGeom(shape[0], shape[1], shape[2:], k, s, p)
    y2 = torch.empty(shape[0], shape[1], *g.odim, device="cuda")
    i2 = torch.empty(shape[0], shape[1], *g.odim, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd(g, zd, y2, i2)                               # two passes
    y1 = torch.empty_like(y2)
    i1 = torch.empty_like(i2)
    ops.maxpool_fwd(g, dev(yraw), y1, i1, in_scale=dev(scale), in_shift=dev(shift), in_relu=True)
    assert torch.equal(y1, y2) and torch.equal(i1, i2)
    ref, _ = F.max_pool3d(z, k, s, p, return_indices=True)
    close(y1, ref, rtol=1e-6, what="fused bn+relu+pool")
    ops.maxpool_fwd(g, dev(yraw), y1, None, in_scale=dev(scale), in_shift=dev(shift), in_relu=False)
    ref2 = F.max_pool3d(yraw * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1), k, s, p)
    close(y1, ref2, rtol=1e-6, what="fused affine+pool")


@pytest.mark.parametrize("k,s,p,shape", [((1, 3, 3), (1, 2, 2), (0, 1, 1), (3, 8, 4, 14, 14)),
                                         ((3, 3, 3), (2, 2, 2), (1, 1, 1), (2, 5, 6, 12, 10)),
                                         ((1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 64, 2, 56, 56)),
                                         ((2, 2, 2), (2, 2, 2), (0, 0, 0), (2, 6, 4, 8, 8))])
@pytest.mark.parametrize("relu,training", [(True, True), (False, True), (True, False)])
def test_batchnorm_backward_through_fused_pool(k, s, p, shape, relu, training):
    """Backward of BatchNorm(+ReLU) -> max-pool with the pool's gradient taken in scattered form
    (never materialising d(activation)): equals autograd of the three ATen operators in float64
    (backbone/s3dg.py:60-64 followed by :151), and is BIT-identical to the product's own separate
    pool-backward + BatchNorm-backward calls on everything but re-associated channel sums."""
    from coclr_amd import ops
    torch.manual_seed(17)
    N, C_ = shape[0], shape[1]
    yraw = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
    gamma = torch.randn(C_, dtype=torch.float64, requires_grad=True)      # both signs
    beta = (torch.randn(C_, dtype=torch.float64) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(C_, dtype=torch.float64) * 0.1, torch.rand(C_, dtype=torch.float64) + 0.5
    z = F.batch_norm(yraw, None if training else rm, None if training else rv, gamma, beta, training,
                     0.1, 1e-3)
    if relu:
        z = torch.relu(z)
    ref = F.max_pool3d(z, k, s, p)
    dyp = torch.randn_like(ref)
    ref.backward(dyp)
    if training:
        mean = yraw.detach().mean((0, 2, 3, 4))
        invstd = (yraw.detach().var((0, 2, 3, 4), unbiased=False) + 1e-3).rsqrt()
    else:
        mean, invstd = rm, (rv + 1e-3).rsqrt()
    scale = gamma.detach() * invstd
    shift = beta.detach() - mean * scale
    f = lambda t: dev(t.float())
    g = ops.PoolGeom(N, C_, shape[2:], k, s, p)
    yd = f(yraw.detach())
    pool_y = torch.empty(N, C_, *g.odim, device="cuda")
    idx = torch.empty(N, C_, *g.odim, dtype=torch.int32, device="cuda")
    ops.maxpool_fwd(g, yd, pool_y, idx, in_scale=f(scale), in_shift=f(shift), in_relu=relu)
    assert ops.pooled_backward_fits(g)
    sums = torch.empty(ops.bn_backward_workspace(N, C_), dtype=torch.float64, device="cuda")
    dy = torch.full(shape, float("nan"), device="cuda")
    dg, db = torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
    ops.bn_act_backward_pooled(g, f(dyp), idx, yd, f(scale), f(shift), f(mean), f(invstd), sums, dy,
                               dg, db, relu, training)
    close(dy, yraw.grad.float(), rtol=2e-5, what="dy through fused pool")
    close(dg, gamma.grad.float(), rtol=2e-5, what="dgamma")
    close(db, beta.grad.float(), rtol=2e-5, what="dbeta")
    # the separate calls
    dz = torch.empty(shape, device="cuda")
    ops.maxpool_bwd(g, f(dyp), idx, dz)
    dy2 = torch.empty(shape, device="cuda")
    dg2, db2 = torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
    ops.bn_act_backward(dz, yd, None, f(scale), f(shift), f(mean), f(invstd), sums, dy2, None, dg2, db2,
                        relu, training)
    close(dy, dy2, rtol=1e-6, what="fused vs separate dy")
    close(dg, dg2, rtol=1e-6, what="fused vs separate dgamma")
    # run-to-run bit-identical (no atomics anywhere in the pair of kernels)
    dy3 = torch.empty(shape, device="cuda")
    ops.bn_act_backward_pooled(g, f(dyp), idx, yd, f(scale), f(shift), f(mean), f(invstd), sums, dy3,
                               dg2, db2, relu, training)
    assert torch.equal(dy, dy3) and torch.equal(dg, dg2) and torch.equal(db, db2)


def test_global_avgpool():
    from coclr_amd import ops
    x = torch.randn(3, 10, 2, 4, 4)
    y = torch.empty(3, 10, 1, 1, 1, device="cuda")
    ops.global_avgpool_fwd(dev(x), y)
    close(y, x.mean((2, 3, 4), keepdim=True), what="avgpool")
    dy = torch.randn(3, 10, 1, 1, 1)
    dx = torch.empty(3, 10, 2, 4, 4, device="cuda")
    ops.global_avgpool_bwd(dev(dy), dx)
    close(dx, (dy / 32).expand(3, 10, 2, 4, 4), what="avgpool bwd")


@pytest.mark.parametrize("M,N,K,ta,tb,splits", [(32, 1024, 1024, False, True, 1),
                                               (4, 128, 1000, False, False, 1),
                                               (37, 200, 70, True, False, 1),
                                               (33, 130, 515, True, True, 4),
                                               (32, 128, 16384, False, True, 128)])
def test_gemm_layouts(M, N, K, ta, tb, splits):
    from coclr_amd import ops
    torch.manual_seed(7)
    A = torch.randn(K, M).t() if ta else torch.randn(M, K)      # logical (M,K)
    Bm = torch.randn(N, K).t() if tb else torch.randn(K, N)     # logical (K,N)
    bias = torch.randn(N)
    ref = torch.relu(0.5 * (A.double() @ Bm.double()).float() + bias)
    Ad = dev(A.t().contiguous()).t() if ta else dev(A.contiguous())
    Bd = dev(Bm.t().contiguous()).t() if tb else dev(Bm.contiguous())
    c = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(1, ops.gemm_workspace(M, N, K, splits)), device="cuda")
    ops.gemm(Ad, Ad.stride(0), Ad.stride(1), Bd, Bd.stride(0), Bd.stride(1), c, N, dev(bias), M, N,
             K, alpha=0.5, relu=True, splits=splits, workspace=ws)
    close(c, ref, what="gemm")


@pytest.mark.parametrize("M,N,K,splits", [(32, 1024, 1024, 32), (8, 1024, 128, 4), (32, 128, 1024, 32),
                                          (5, 100, 70, 2), (32, 2048, 2048, 16)])
def test_gemm_fused_fold_is_the_two_launch_fold(M, N, K, splits):
    """coclr_gemm_fused folds its split-K partials in the order coclr_gemm does: bit-identical, run after run."""
    from coclr_amd import ops
    torch.manual_seed(17)
    A, Bm, bias = dev(torch.randn(M, K)), dev(torch.randn(N, K)), dev(torch.randn(N))
    ref = torch.empty(M, N, device="cuda")
    ws = torch.empty(max(1, ops.gemm_workspace(M, N, K, splits)), device="cuda")
    ops.gemm(A, K, 1, Bm, 1, K, ref, N, bias, M, N, K, alpha=0.5, relu=True, splits=splits, workspace=ws)
    for _ in range(3):
        got = torch.full((M, N), float("nan"), device="cuda")
        ws2 = torch.empty(ops.gemm_fused_workspace(M, N, K, splits), device="cuda")
        ops.gemm_fused(A, K, 1, Bm, 1, K, got, N, bias, M, N, K, alpha=0.5, relu=True, splits=splits,
                       workspace=ws2)
        assert torch.equal(got, ref)
    close(ref, torch.relu(0.5 * (A.double() @ Bm.double().t()).float() + bias), what="gemm")


def test_gemm_fused_epilogues():
    """The row-level epilogues of the projection head (model/pretrain.py:49-54,153-154,175-182 and their
    backward) against ATen: ReLU backward, F.normalize, l_pos + F.normalize backward, average-pool backward,
    and the bias gradient as the row sums of a weight-gradient product."""
    from coclr_amd import ops
    torch.manual_seed(18)
    B, D, C_, K, T, S = 6, 128, 200, 640, 0.07, 12

    def fused(A, Bkn, M, N, Kd, splits, c=None, **kw):
        c = torch.empty(M, N, device="cuda") if c is None else c
        ws = torch.empty(ops.gemm_fused_workspace(M, N, Kd, splits), device="cuda")
        ops.gemm_fused(A, A.stride(0), A.stride(1), Bkn, Bkn.stride(0), Bkn.stride(1), c, N, None, M, N, Kd,
                       splits=splits, workspace=ws, **kw)
        return c

    # mode 1: (dy . W) masked by the ReLU's output
    dy, W, h = torch.randn(B, D), torch.randn(D, C_), torch.randn(B, C_)
    got = fused(dev(dy), dev(W), B, C_, D, 4, mode=1, ep_a=dev(h), lda=C_)
    close(got, (dy @ W) * (h > 0), what="relu backward epilogue")
    # mode 2: F.normalize of the product's rows
    x, W2 = torch.randn(B, C_), torch.randn(D, C_)
    inv = torch.empty(B, device="cuda")
    got = fused(dev(x), dev(W2).t(), B, D, C_, 3, mode=2, out2=inv, f=1e-12)
    y = x @ W2.t()
    close(got, F.normalize(y, dim=1), what="normalize epilogue")
    close(inv, 1 / y.norm(dim=1), what="inverse norms")
    # mode 3: logits backward (queue term + l_pos term) through F.normalize
    xq = torch.randn(B, D, requires_grad=True)
    q = F.normalize(xq, dim=1)
    k = F.normalize(torch.randn(B, D), dim=1)
    queue = F.normalize(torch.randn(D, K), dim=0)
    logits = torch.cat([(q * k).sum(1, keepdim=True), q @ queue], 1) / T
    dl = torch.randn_like(logits)
    logits.backward(dl)
    dld = dev(dl)
    invq = dev(1 / xq.detach().norm(dim=1))
    got = torch.empty(B, D, device="cuda")
    ws = torch.empty(ops.gemm_fused_workspace(B, D, K, 5), device="cuda")
    queued = dev(queue)
    ops.gemm_fused(dld[:, 1:], 1 + K, 1, queued, 1, K, got, D, None, B, D, K, alpha=1 / T, splits=5, workspace=ws,
                   mode=3, ep_a=dld, lda=1 + K, ep_b=dev(k), ep_y=dev(q.detach()), inv_norm=invq, f=1 / T)
    close(got, xq.grad, what="logits + normalize backward epilogue")
    # mode 4: the product spread over planes of S positions
    dh, W1 = torch.randn(B, D), torch.randn(D, C_)
    dx = torch.empty(B, C_, 2, 3, 2, device="cuda")
    fused(dev(dh), dev(W1), B, C_, D, 2, c=dx, mode=4, S=S)
    close(dx, ((dh @ W1) / S)[:, :, None, None, None].expand(B, C_, 2, 3, 2), what="average-pool backward epilogue")
    # weight gradient + bias gradient (row sums of A) in one launch, no split
    dyo, xin = torch.randn(B, D), torch.randn(B, C_)
    dw, db = torch.empty(D, C_, device="cuda"), torch.empty(D, device="cuda")
    dyd, xd = dev(dyo), dev(xin)
    ops.gemm_fused(dyd, 1, D, xd, C_, 1, dw, C_, None, D, C_, B, rowsum=db)
    close(dw, dyo.t() @ xin, what="weight gradient")
    close(db, dyo.sum(0), what="bias gradient")


def test_l2norm_and_logits():
    from coclr_amd import ops
    torch.manual_seed(8)
    B, D, K, T = 6, 128, 640, 0.07
    x = torch.randn(B, D, requires_grad=True)
    q = F.normalize(x, dim=1)
    k = F.normalize(torch.randn(B, D), dim=1)
    queue = F.normalize(torch.randn(D, K), dim=0)
    logits = torch.cat([torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1),
                        torch.einsum('nc,ck->nk', [q, queue])], 1) / T
    dl = torch.randn_like(logits)
    logits.backward(dl)

    xd = dev(x.detach())
    qd, inv = torch.empty_like(xd), torch.empty(B, device="cuda")
    ops.l2norm_fwd(xd, qd, inv)
    close(qd, q, what="normalize")
    lg = torch.empty(B, 1 + K, device="cuda")
    ops.nce_logits_fwd(qd, dev(k), dev(queue), lg, T)
    close(lg, logits, what="logits")
    dq = torch.empty(B, D, device="cuda")
    splits = 5
    ws = torch.empty(ops.gemm_workspace(B, D, K, splits), device="cuda")
    ops.nce_logits_bwd(dev(dl), dev(k), dev(queue), dq, ws, T, splits)
    dx = torch.empty_like(xd)
    ops.l2norm_bwd(dq, qd, inv, dx)
    close(dx, x.grad, rtol=5e-4, what="normalize+logits bwd")


def test_momentum_enqueue_mask_gather():
    from coclr_amd import ops
    torch.manual_seed(9)
    # momentum over two tensors, one spanning several chunks
    pk = [torch.randn(70000), torch.randn(33)]
    pq = [torch.randn(70000), torch.randn(33)]
    pkd, pqd = [dev(t) for t in pk], [dev(t) for t in pq]
    rows = []
    for a, b in zip(pkd, pqd):
        for off in range(0, a.numel(), 32768):
            rows.append((a.data_ptr() + 4 * off, b.data_ptr() + 4 * off, min(32768, a.numel() - off)))
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    m = 0.999
    ops.momentum_update(table, len(rows), float(m), float(1. - m))
    for a, b, c in zip(pkd, pk, pq):
        assert torch.equal(a.cpu(), b * m + c * (1. - m)), "momentum must be bit-exact"

    # enqueue with wrap-around
    D, K, BW = 16, 12, 4
    queue = torch.randn(D, K)
    qd = dev(queue)
    ptr = torch.tensor([8], dtype=torch.long).cuda()
    keys = torch.randn(BW, D)
    lab = torch.full((K,), -1, dtype=torch.long).cuda()
    ops.queue_enqueue(qd, dev(keys), ptr)
    ops.queue_fill_i64(lab, torch.arange(BW).cuda(), 0, BW, ptr)
    ops.queue_advance(ptr, BW, K)
    queue[:, 8:12] = keys.T
    assert torch.equal(qd.cpu(), queue) and int(ptr) == 0
    assert lab.cpu().tolist() == [-1] * 8 + [0, 1, 2, 3]
    ops.queue_fill_i64(lab, None, 1, BW, ptr)
    assert lab.cpu().tolist()[:4] == [1, 1, 1, 1]

    # positive mask + top-k mining
    B, K = 5, 300
    sim = torch.randn(B, K)
    src = torch.randint(0, 4, (B,))
    names = torch.randint(0, 4, (K,))
    names[:7] = -1
    mask_source = src[:, None] == names[None, :]
    ms = sim.clone()
    ms[mask_source] = -float("inf")
    _, idx = torch.topk(ms, 5, dim=1)
    exp = mask_source.clone()
    exp.scatter_(1, idx, True)
    exp = torch.cat([torch.ones(B, 1, dtype=torch.bool), exp], 1)
    mask = torch.empty(B, 1 + K, dtype=torch.uint8, device="cuda")
    ops.positive_mask(dev(sim), dev(src), dev(names), mask, 5)
    assert torch.equal(mask.cpu().bool(), exp)
    ops.positive_mask(None, dev(src), dev(names), mask, 0)
    assert torch.equal(mask.cpu().bool()[:, 1:], mask_source)

    # row gather
    x = torch.randn(6, 3, 4, 5)
    idx = torch.tensor([5, 0, 3])
    out = torch.empty(3, 3, 4, 5, device="cuda")
    ops.gather_rows(dev(x), dev(idx), out)
    assert torch.equal(out.cpu(), x[idx])

    # relu + colsum
    y = torch.empty(6, 60, device="cuda")
    xr = torch.randn(6, 60)
    ops.relu_fwd(dev(xr), y)
    assert torch.equal(y.cpu(), torch.relu(xr))
    cs = torch.empty(60, device="cuda")
    ops.colsum(dev(xr), cs)
    close(cs, xr.sum(0), what="colsum")


def test_self_gating_kernels():
    """sigmoid / per-plane scale / per-plane dot of S3D-G's SelfGating (backbone/s3dg.py:68-78),
    with channel-slice operands and the accumulate form used by the backward."""
    from coclr_amd import ops
    torch.manual_seed(12)
    N, Cc, dims = 3, 10, (2, 5, 7)
    wide = torch.randn(N, Cc + 4, *dims)
    a = wide[:, 2:2 + Cc]
    gain, bias = torch.randn(N, Cc), torch.randn(N, Cc)
    wd = dev(wide)
    out = torch.zeros(N, Cc + 6, *dims, device="cuda")
    ops.plane_scale(wd[:, 2:2 + Cc], dev(gain), None, out[:, 3:3 + Cc])
    close(out[:, 3:3 + Cc], a * gain[:, :, None, None, None], what="plane_scale")
    assert out[:, :3].abs().max().item() == 0 and out[:, 3 + Cc:].abs().max().item() == 0
    base = torch.randn(N, Cc, *dims)
    acc = dev(base)
    ops.plane_scale(wd[:, 2:2 + Cc], dev(gain), dev(bias), acc, accumulate=True)
    close(acc, base + a * gain[:, :, None, None, None] + bias[:, :, None, None, None],
          what="plane_scale accumulate")
    b = torch.randn(N, Cc, *dims)
    dots = torch.empty(N, Cc, device="cuda")
    ops.plane_dot(wd[:, 2:2 + Cc], dev(b), dots)
    close(dots, (a * b).sum((2, 3, 4)), what="plane_dot")
    s_ = torch.randn(N, Cc) * 3
    w = torch.empty(N, Cc, device="cuda")
    ops.sigmoid_fwd(dev(s_), w)
    close(w, torch.sigmoid(s_), what="sigmoid")
    dw = torch.randn(N, Cc)
    ds = torch.empty(N, Cc, device="cuda")
    ops.sigmoid_bwd(dev(dw), w, ds)
    sg = torch.sigmoid(s_)
    close(ds, dw * sg * (1 - sg), what="sigmoid backward")


WINO_CASES = [(2, 192, 192, (8, 8, 8)), (3, 208, 208, (4, 8, 8)), (2, 48, 48, (2, 4, 4)),
              (2, 64, 64, (3, 7, 7)), (2, 64, 96, (5, 6, 10)), (4, 32, 24, (16, 4, 4)),
              (2, 128, 128, (1, 4, 4)), (3, 96, 80, (16, 16, 16)), (5, 70, 130, (7, 3, 5))]


@pytest.mark.parametrize("algo", [1, 2], ids=["F23", "F43"])
@pytest.mark.parametrize("case", WINO_CASES, ids=lambda c: "%d_%d_%d_%s" % c)
def test_conv_temporal_winograd(case, algo):
    """(3,1,1) stride-1 pad-1 convolutions through Winograd along T -- F(2,3) over frame pairs (algo = 1) and
    F(4,3) over frame quads (algo = 2, what conv_geom() picks from four frames up): forward with BatchNorm
    partial sums, accumulate form, the inference epilogue (affine + ReLU: CoCLR's frozen sampler) and the data
    gradient -- frame counts that are not multiples of the group, ragged channel counts, a single frame."""
    from coclr_amd import ops, engine
    N, Cin, Cout, dims = case
    k, s, p = (3, 1, 1), (1, 1, 1), (1, 0, 0)
    torch.manual_seed(6)
    x = torch.randn(N, Cin, *dims, requires_grad=True)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, s, p)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    picked = ops.conv_geom(N, Cin, Cout, dims, k, s, p)
    assert picked.algo == (2 if ops.winograd_t4_pays(dims) else 1)
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=algo)
    assert g.dgrad().algo == algo
    run = engine.Run(torch.device("cuda"), save=False)
    wd, xd, dyd = dev(w.detach()), dev(x.detach()), dev(dy)
    y = torch.full((N, Cout, *g.odim), float("nan"), device="cuda")
    stats = torch.empty(2 * Cout * g.ntiles(), device="cuda")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=algo), y, stats=stats)
    close(y, ref, what="winograd fwd")
    st = stats.view(2, Cout, -1).double().sum(-1).cpu()
    close(st[0], ref.double().sum((0, 2, 3, 4)), rtol=1e-3, what="stats sum")
    close(st[1], (ref.double() ** 2).sum((0, 2, 3, 4)), what="stats sumsq")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=algo), y, accumulate=True)
    close(y, 2 * ref, what="winograd accumulate")
    sc, sf = torch.rand(Cout) + 0.5, torch.randn(Cout)
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=algo), y, ep_scale=dev(sc), ep_shift=dev(sf), relu=True)
    close(y, torch.relu(ref.detach() * sc.view(1, -1, 1, 1, 1) + sf.view(1, -1, 1, 1, 1)),
          what="winograd affine+relu epilogue")
    dx = torch.full((N, Cin, *dims), float("nan"), device="cuda")
    dg = g.dgrad()
    ops.conv_fwd(dg, dyd, run.pack(wd, True, algo=algo), dx)
    close(dx, x.grad, what="winograd dgrad")
    # weight gradient: with desc.algo = 1 the wide layers (>= 48 channels both ways) take the
    # Winograd F(2,3) form (four MFMAs per frame pair instead of six), the others the direct one
    dw = torch.full_like(wd, float("nan"))
    ws = torch.empty(g.wgrad_workspace(), device="cuda")
    ops.conv_wgrad(g, xd, dyd, dw, ws, Cin * 3, 3, 0)
    close(dw, w.grad, what="winograd wgrad")
    ops.conv_wgrad(g, xd, dyd, dw, ws, Cin * 3, 3, 0, accumulate=True)
    close(dw, 2 * w.grad, what="winograd wgrad accumulate")
    # the same through a channel-slice view of a wider dY (concat-free inception gradients)
    wide = torch.zeros(N, Cout + 5, *dims, device="cuda")
    wide[:, 3:3 + Cout] = dyd
    ops.conv_wgrad(g, xd, wide[:, 3:3 + Cout], dw, ws, Cin * 3, 3, 0)
    close(dw, w.grad, what="winograd wgrad from a dY slice")


POLY7_CASES = [(2, 64, 64, (32, 16, 16)), (3, 64, 64, (16, 8, 8)), (2, 48, 72, (18, 5, 7)), (5, 16, 130, (16, 4, 4)),
               (2, 96, 64, (20, 12, 12)), (32, 64, 64, (16, 8, 8))]


@pytest.mark.parametrize("case", POLY7_CASES, ids=lambda c: "%d_%d_%d_%s" % c)
def test_conv_temporal_stem_polyphase_winograd(case):
    """The (7,1,1) / stride (2,1,1) / pad (3,0,0) temporal stem conv (backbone/s3dg.py:41,145) in polyphase
    Winograd form (algo = 1 on that stencil: F(2,3) on the odd taps + F(2,4) on the even ones, nine channel
    contractions per pair of output frames instead of fourteen) against F.conv3d and against the direct kernel:
    forward with BatchNorm partial sums, accumulate form, inference epilogue; odd output frame counts, ragged
    channels.  The data gradient of this conv runs in phases (ConvGeom.dgrad_phases), not here."""
    from coclr_amd import ops, engine, _lib
    N, Cin, Cout, dims = case
    k, s, p = (7, 1, 1), (2, 1, 1), (3, 0, 0)
    torch.manual_seed(16)
    x = torch.randn(N, Cin, *dims)
    w = torch.randn(Cout, Cin, *k) * 0.05
    ref = F.conv3d(x, w, None, s, p)
    picked = ops.conv_geom(N, Cin, Cout, dims, k, s, p)
    assert picked.algo == (1 if (ops.WINOGRAD_POLY7 and dims[0] >= 32) else 0) and picked.dgrad().algo == 0
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=1)
    g0 = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=0)
    with pytest.raises(ValueError):
        ops.ConvGeom(N, Cin, Cout, (dims[0] + 1,) + dims[1:], k, s, p, algo=1)        # odd frame count
    with pytest.raises(ValueError):
        ops.ConvGeom(N, Cin, Cout, (8,) + dims[1:], k, s, p, algo=1)                  # fewer than four output pairs
    assert ops.conv_geom(N, Cin, Cout, (8,) + dims[1:], k, s, p).algo == 0
    run = engine.Run(torch.device("cuda"), save=False)
    wd, xd = dev(w), dev(x)
    y = torch.full((N, Cout, *g.odim), float("nan"), device="cuda")
    stats = torch.empty(2 * Cout * g.ntiles(), device="cuda")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, stats=stats)
    close(y, ref, what="polyphase fwd")
    y0 = torch.empty_like(y)
    ops.conv_fwd(g0, xd, run.pack(wd, False), y0)
    close(y, y0, rtol=2e-5, what="polyphase vs direct kernel")
    st = stats.view(2, Cout, -1).double().sum(-1).cpu()
    close(st[0], ref.double().sum((0, 2, 3, 4)), rtol=1e-3, what="stats sum")
    close(st[1], (ref.double() ** 2).sum((0, 2, 3, 4)), what="stats sumsq")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, accumulate=True)
    close(y, 2 * ref, what="polyphase accumulate")
    sc, sf = torch.rand(Cout) + 0.5, torch.randn(Cout)
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, ep_scale=dev(sc), ep_shift=dev(sf), relu=True)
    close(y, torch.relu(ref * sc.view(1, -1, 1, 1, 1) + sf.view(1, -1, 1, 1, 1)), what="polyphase affine+relu epilogue")
    # consumer-side BatchNorm apply (coclr_conv_call.in_scale / in_shift): x is the RAW output of the unit in
    # front; zero padding must stay zero AFTER the affine (relu(shift) != 0)
    isc, ish = torch.rand(Cin) + 0.5, torch.randn(Cin)
    for relu_in in (True, False):
        xz = x * isc.view(1, -1, 1, 1, 1) + ish.view(1, -1, 1, 1, 1)
        xz = torch.relu(xz) if relu_in else xz
        refz = F.conv3d(xz, w, None, s, p)
        stz = torch.empty(2 * Cout * g.ntiles(), device="cuda")
        ops.conv_fwd_multi([dict(geom=g, x=xd, w=run.pack(wd, False, algo=1), y=y, stats=stz,
                                 in_affine=(dev(isc), dev(ish), relu_in))])
        close(y, refz, what="polyphase with the producing unit's BatchNorm%s applied on load"
              % ("+ReLU" if relu_in else ""))
        close(stz.view(2, Cout, -1).double().sum(-1).cpu()[1], (refz.double() ** 2).sum((0, 2, 3, 4)),
              what="its statistics")
    with pytest.raises(_lib.HipLibraryError):          # the direct kernel has no operand path for it
        ops.conv_fwd_multi([dict(geom=g0, x=xd, w=run.pack(wd, False), y=y, in_affine=(dev(isc), dev(ish), True))])
    # into a channel slice of a wider tensor, from a channel slice of a wider input
    wide = torch.zeros(N, Cout + 8, *g.odim, device="cuda")
    xw = torch.zeros(N, Cin + 4, *dims, device="cuda")
    xw[:, 2:2 + Cin] = xd
    ops.conv_fwd(g, xw[:, 2:2 + Cin], run.pack(wd, False, algo=1), wide[:, 8:])
    close(wide[:, 8:], ref, what="polyphase between channel slices")
    assert wide[:, :8].abs().max().item() == 0


WINO_HW_CASES = [(2, 64, 192, (2, 32, 32)), (2, 192, 208, (4, 16, 16)), (3, 48, 96, (3, 8, 8)),
                 (2, 32, 24, (5, 4, 4)), (2, 16, 48, (1, 6, 10)), (5, 160, 320, (2, 8, 8)),
                 (2, 24, 64, (2, 12, 20)), (9, 40, 64, (3, 4, 4)),
                 (33, 64, 64, (2, 16, 16))]            # more tiles than persistent workgroups: the tile loop


@pytest.mark.parametrize("waves", ["two_per_simd", "one_per_simd"])
@pytest.mark.parametrize("case", WINO_HW_CASES, ids=lambda c: "%d_%d_%d_%s" % c)
def test_conv_spatial_winograd(case, waves, monkeypatch, algo=1):
    """(1,3,3) stride-1 pad-1 convolutions through Winograd F(2x2,3x3) (algo = 1): forward with
    BatchNorm partial sums, accumulate form, fused affine+ReLU epilogue, and the data gradient --
    ragged channel counts, maps that are not a power of two, boxes spanning frames and samples.
    Both kernels: conv_wino_hw8_kernel (two waves per SIMD; rows of whole 16-byte granules and channel
    counts that are multiples of 8, everything else falls through) and conv_wino_hw_kernel."""
    from coclr_amd import ops, engine
    monkeypatch.setenv("COCLR_WINO_W8", "1" if waves == "two_per_simd" else "0")
    N, Cin, Cout, dims = case
    k, s, p = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    torch.manual_seed(7)
    x = torch.randn(N, Cin, *dims, requires_grad=True)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, s, p)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=algo)   # conv_geom() keeps small maps direct
    assert g.dgrad().algo == algo
    assert ops.conv_geom(N, Cin, Cout, dims, k, s, p).algo == (1 if min(dims[1:]) >= 16 else 0)
    run = engine.Run(torch.device("cuda"), save=False)
    wd, xd, dyd = dev(w.detach()), dev(x.detach()), dev(dy)
    y = torch.full((N, Cout, *g.odim), float("nan"), device="cuda")
    stats = torch.empty(2 * Cout * g.ntiles(), device="cuda")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, stats=stats)
    close(y, ref, what="winograd fwd")
    st = stats.view(2, Cout, -1).double().sum(-1).cpu()
    close(st[0], ref.double().sum((0, 2, 3, 4)), rtol=1e-3, what="stats sum")
    close(st[1], (ref.double() ** 2).sum((0, 2, 3, 4)), what="stats sumsq")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, accumulate=True)
    close(y, 2 * ref, what="winograd accumulate")
    sc, sf = torch.rand(Cout) + 0.5, torch.randn(Cout)
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), y, ep_scale=dev(sc), ep_shift=dev(sf), relu=True)
    close(y, torch.relu(ref * sc[None, :, None, None, None] + sf[None, :, None, None, None]),
          what="winograd affine+relu epilogue")
    dx = torch.full((N, Cin, *dims), float("nan"), device="cuda")
    dg = g.dgrad()
    ops.conv_fwd(dg, dyd, run.pack(wd, True, algo=1), dx)
    close(dx, x.grad, what="winograd dgrad")
    # a channel slice of a wider destination (concat-free inception output)
    wide = torch.zeros(N, Cout + 8, *g.odim, device="cuda")
    ops.conv_fwd(g, xd, run.pack(wd, False, algo=1), wide[:, 8:])
    close(wide[:, 8:], ref, what="winograd into a channel slice")
    assert float(wide[:, :8].abs().max()) == 0.0



def test_pack_batch_matches_single_launches():
    """coclr_conv_pack_describe + coclr_conv_pack_batch re-lay a mixed set of operands (forward,
    data-gradient, tap subsets, both Winograd forms, sub-blocks placed in a concatenated operand)
    bit-identically to one coclr_conv_pack_weights launch each."""
    from coclr_amd import ops
    torch.manual_seed(11)
    reqs = []     # (w, size, args, kw)

    def add(w, taps_packed, transpose, args):
        cout, cin = w.shape[:2]
        reqs.append((w, ops.conv_packed_size(cin, cout, taps_packed, transpose), args, {}))

    w133 = dev(torch.randn(208, 48, 1, 3, 3))
    w311 = dev(torch.randn(96, 64, 3, 1, 1))
    w711 = dev(torch.randn(64, 64, 7, 1, 1))
    w111a, w111b = dev(torch.randn(40, 72, 1, 1, 1)), dev(torch.randn(24, 72, 1, 1, 1))
    add(w133, 9, 0, (208, 48, 9, 48 * 9, 9, 0, 0, 1))
    add(w133, 9, 1, (208, 48, 9, 48 * 9, 9, 0, 1, 1))
    add(w133, 16, 0, (208, 48, 16, 48 * 9, 9, 0, 2, 1))
    add(w133, 16, 1, (208, 48, 16, 48 * 9, 9, 0, 3, 1))
    add(w311, 4, 0, (96, 64, 4, 64 * 3, 3, 0, 2, 1))
    add(w311, 4, 1, (96, 64, 4, 64 * 3, 3, 0, 3, 1))
    add(w711, 4, 1, (64, 64, 4, 64 * 7, 7, 0, 1, 2))          # taps 0,2,4,6: one dgrad phase
    add(w711, 3, 1, (64, 64, 3, 64 * 7, 7, 1, 1, 2))          # taps 1,3,5
    ncat = ops.conv_packed_size(72, 64, 1, 0)
    reqs.append((w111a, ncat, (40, 72, 1, 72, 1, 0, 0, 1), dict(row0=0, rows_total=72, col0=0, cols_total=64)))
    reqs.append((w111b, ncat, (24, 72, 1, 72, 1, 0, 0, 1), dict(row0=0, rows_total=72, col0=40, cols_total=64)))
    single = [torch.zeros(n, device="cuda") for _, n, _, _ in reqs]
    batch = [torch.zeros(n, device="cuda") for _, n, _, _ in reqs]
    batch[-1] = batch[-2]; single[-1] = single[-2]            # the two heads share one operand
    rows, bmap = [], []
    for i, (w, n, args, kw) in enumerate(reqs):
        ops.conv_pack_weights(w, single[i], *args, **kw)
        row, nb = ops.conv_pack_describe(w, batch[i], *args, **kw)
        assert nb == -(-_pack_elems(args, kw) // 1024)
        rows.append(row)
        bmap.extend((i, b) for b in range(nb))
    table = torch.tensor(rows, dtype=torch.int64).cuda()
    blockmap = torch.tensor(bmap, dtype=torch.int32).cuda()
    ops.conv_pack_batch(table, blockmap)
    for i in range(len(reqs)):
        assert torch.equal(single[i], batch[i]), "operand %d differs" % i
    assert float(single[2].abs().sum()) > 0


def _pack_elems(args, kw):
    cout, cin, taps = args[0], args[1], args[2]
    transpose = args[6] & 1
    r, c = (cout, cin) if transpose else (cin, cout)
    if kw:
        return taps * r * c
    return taps * ((r + 31) // 32 * 32) * ((c + 127) // 128 * 128)


WINO_HW_WGRAD_CASES = [(2, 64, 192, (2, 32, 32)), (2, 192, 208, (4, 16, 16)), (3, 48, 96, (3, 8, 8)),
                       (5, 160, 320, (2, 8, 8)), (2, 96, 128, (3, 16, 16)), (2, 130, 70, (2, 12, 20)),
                       (3, 64, 50, (1, 4, 4)), (32, 64, 192, (1, 32, 32))]


@pytest.mark.parametrize("case", WINO_HW_WGRAD_CASES, ids=lambda c: "%d_%d_%d_%s" % c)
def test_conv_spatial_winograd_weight_gradient(case, monkeypatch):
    """Weight gradient of the wide (1,3,3) layers in the F(2x2,3x3) domain (desc.algo = 1, >= 48 channels
    both ways: 16 MFMAs per two 2x2 blocks instead of 36, ABI 15): against ATen's fp32 CPU gradient, the
    accumulate form, a dY channel slice, run-to-run bit-identity, and against the direct kernel on the
    same data (algo = 0)."""
    from coclr_amd import ops
    N, Cin, Cout, dims = case
    k, s, p = (1, 3, 3), (1, 1, 1), (0, 1, 1)
    torch.manual_seed(17)
    x = torch.randn(N, Cin, *dims)
    w = (torch.randn(Cout, Cin, *k) * 0.05).requires_grad_(True)
    ref = F.conv3d(x, w, None, s, p)
    dy = torch.randn_like(ref)
    ref.backward(dy)
    g = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=1)
    g0 = ops.ConvGeom(N, Cin, Cout, dims, k, s, p, algo=0)
    # two split slices per split in the Winograd form (one per xi-half): the planner reports them
    assert g.wgrad_workspace() % (Cout * Cin * 9) == 0
    xd, dyd = dev(x), dev(dy)
    dw = torch.full((Cout, Cin, *k), float("nan"), device="cuda")
    ws = torch.empty(g.wgrad_workspace(), device="cuda")
    ops.conv_wgrad(g, xd, dyd, dw, ws, Cin * 9, 9, 0)
    close(dw, w.grad, what="winograd-domain wgrad")
    dw0 = torch.empty_like(dw)
    ws0 = torch.empty(g0.wgrad_workspace(), device="cuda")
    ops.conv_wgrad(g0, xd, dyd, dw0, ws0, Cin * 9, 9, 0)
    close(dw, dw0, rtol=2e-5, what="winograd-domain vs direct wgrad")
    ops.conv_wgrad(g, xd, dyd, dw, ws, Cin * 9, 9, 0, accumulate=True)
    close(dw, 2 * w.grad, what="winograd-domain wgrad accumulate")
    wide = torch.zeros(N, Cout + 5, *dims, device="cuda")
    wide[:, 3:3 + Cout] = dyd
    ops.conv_wgrad(g, xd, wide[:, 3:3 + Cout], dw, ws, Cin * 9, 9, 0)
    close(dw, w.grad, what="winograd-domain wgrad from a dY slice")
    # run-to-run determinism (fixed fold order of the split slices)
    dw2 = torch.empty_like(dw)
    ops.conv_wgrad(g, xd, wide[:, 3:3 + Cout], dw2, ws, Cin * 9, 9, 0)
    assert torch.equal(dw, dw2)
