"""Multi-problem launches (ABI 15, sums epilogue: 16): coclr_conv3d_fwd_multi, coclr_bn_finalize_apply_multi,
coclr_bn_act_backward_multi and the engine's lockstep emission of the two separable branches of an
inception block (backbone/s3dg.py:100-118).  Every problem keeps the plan it has alone, so the bar is
BIT-IDENTITY with the single launches: outputs, BatchNorm statistics, running buffers, gradients."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

B = 32


def _conv_case(run, cin, cout, k, p, idim, seed, dgrad=False, accumulate=False, stats=True, algo=None):
    from coclr_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    geom = ops.conv_geom(B, cin, cout, idim, k, (1, 1, 1), p)
    if algo is not None and geom.algo != algo:
        geom = ops.ConvGeom(B, cin, cout, idim, k, (1, 1, 1), p, algo=algo)
    w = torch.randn(cout, cin, *k, device="cuda", generator=g) * 0.05
    if dgrad:
        dg = geom.dgrad()
        x = torch.randn(B, cout, *geom.odim, device="cuda", generator=g)
        y0 = torch.randn(B, cin, *idim, device="cuda", generator=g)
        return dict(geom=dg, x=x, w=run.pack(w, True, algo=dg.algo), y0=y0, accumulate=accumulate,
                    stats=False, keep=w)
    x = torch.randn(B, cin, *idim, device="cuda", generator=g)
    y0 = torch.randn(B, cout, *geom.odim, device="cuda", generator=g)
    return dict(geom=geom, x=x, w=run.pack(w, False, algo=geom.algo), y0=y0, accumulate=False, stats=stats,
                keep=w)


def _run_single(case):
    from coclr_amd import ops
    y = case["y0"].clone()
    st = torch.full((2 * case["geom"].Cout * case["geom"].ntiles(),), 7.0, device="cuda") if case["stats"] else None
    ops.conv_fwd(case["geom"], case["x"], case["w"], y, stats=st, accumulate=case["accumulate"])
    return y, st


PAIRS = [
    # (1,3,3) direct, the wide branch on 64x128 tiles and the narrow one on 64x64: mixed-variant launch
    ("4b conv1", (96, 208, (1, 3, 3), (0, 1, 1), (8, 8, 8)), (16, 48, (1, 3, 3), (0, 1, 1), (8, 8, 8))),
    # same variant
    ("4f conv1", (160, 320, (1, 3, 3), (0, 1, 1), (8, 8, 8)), (32, 128, (1, 3, 3), (0, 1, 1), (8, 8, 8))),
    ("5c conv1", (192, 384, (1, 3, 3), (0, 1, 1), (4, 4, 4)), (48, 128, (1, 3, 3), (0, 1, 1), (4, 4, 4))),
    # temporal Winograd
    ("4b conv2", (208, 208, (3, 1, 1), (1, 0, 0), (8, 8, 8)), (48, 48, (3, 1, 1), (1, 0, 0), (8, 8, 8))),
    ("5c conv2", (384, 384, (3, 1, 1), (1, 0, 0), (4, 4, 4)), (128, 128, (3, 1, 1), (1, 0, 0), (4, 4, 4))),
    # 16x16x16 maps: spatial Winograd (no pair kernel: two launches inside the call) and temporal pairs
    ("3b conv1", (96, 128, (1, 3, 3), (0, 1, 1), (16, 16, 16)), (16, 32, (1, 3, 3), (0, 1, 1), (16, 16, 16))),
    ("3b conv2", (128, 128, (3, 1, 1), (1, 0, 0), (16, 16, 16)), (32, 32, (3, 1, 1), (1, 0, 0), (16, 16, 16))),
    # pointwise: the fused heads of a block beside the pool branch's convolution
    ("4b heads+b3", (480, 304, (1, 1, 1), (0, 0, 0), (8, 8, 8)), (480, 64, (1, 1, 1), (0, 0, 0), (8, 8, 8))),
    ("5c heads+b3", (832, 624, (1, 1, 1), (0, 0, 0), (4, 4, 4)), (832, 128, (1, 1, 1), (0, 0, 0), (4, 4, 4))),
    ("3c heads+b3", (256, 288, (1, 1, 1), (0, 0, 0), (16, 16, 16)), (256, 64, (1, 1, 1), (0, 0, 0), (16, 16, 16))),
    # not the same stencil at all: falls apart into two launches, results unchanged
    ("mismatch", (96, 208, (1, 3, 3), (0, 1, 1), (8, 8, 8)), (480, 64, (1, 1, 1), (0, 0, 0), (8, 8, 8))),
]


@pytest.mark.parametrize("name,a,b", PAIRS, ids=[p[0] for p in PAIRS])
@pytest.mark.parametrize("mode", ["forward", "dgrad", "dgrad_accumulate"])
def test_conv_pairs_are_bit_identical_to_single_launches(name, a, b, mode):
    from coclr_amd import engine, ops
    run = engine.Run(torch.device("cuda"), save=False)
    dgrad = mode != "forward"
    cases = [_conv_case(run, *spec, seed=11 + i, dgrad=dgrad, accumulate=mode == "dgrad_accumulate")
             for i, spec in enumerate((a, b))]
    singles = [_run_single(c) for c in cases]
    ys = [c["y0"].clone() for c in cases]
    sts = [torch.full((2 * c["geom"].Cout * c["geom"].ntiles(),), 7.0, device="cuda") if c["stats"] else None
           for c in cases]
    ops.conv_fwd_multi([dict(geom=c["geom"], x=c["x"], w=c["w"], y=y, stats=st, accumulate=c["accumulate"])
                        for c, y, st in zip(cases, ys, sts)])
    torch.cuda.synchronize()
    for (y1, s1), y2, s2 in zip(singles, ys, sts):
        assert torch.equal(y1, y2), "%s %s: outputs differ" % (name, mode)
        if s1 is not None:
            assert torch.equal(s1, s2), "%s %s: statistics differ" % (name, mode)


def _bn_units(seed, shapes, slices=False):
    """BatchNorm units over freshly drawn tensors; `slices`: the units' y are channel ranges of one wider
    tensor and share one statistics buffer (the fused heads of an inception block)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    units = []
    if slices:
        N, dims = shapes[0][0], shapes[0][2]
        ctot = sum(s[1] for s in shapes)
        ywide = torch.randn(N, ctot, *dims, device="cuda", generator=g)
        ntiles = 16
        stats = torch.rand(2 * ctot * ntiles, device="cuda", generator=g)
    c0 = 0
    for (N, C_, dims) in shapes:
        y = ywide[:, c0:c0 + C_] if slices else torch.randn(N, C_, *dims, device="cuda", generator=g)
        nt = ntiles if slices else 8
        st = stats if slices else torch.rand(2 * C_ * nt, device="cuda", generator=g)
        # partial sums consistent with y would be needed for meaningful statistics; any finite numbers do
        # for an identity check of the arithmetic (sumsq kept large enough for a positive variance)
        if not slices:
            st[C_ * nt:] += 4.0
        units.append(dict(N=N, C=C_, dims=dims, y=y, stats=st, ntiles=nt, c0=c0 if slices else 0,
                          c_total=ctot if slices else None,
                          gamma=torch.rand(C_, device="cuda", generator=g) + 0.5,
                          beta=torch.randn(C_, device="cuda", generator=g)))
        c0 += C_
    if slices:
        stats[ctot * ntiles:] += 4.0
    return units


def _bn_forward(units, multi):
    from coclr_amd import ops
    outs, calls = [], []
    for u in units:
        C_ = u["C"]
        small = torch.zeros(4, C_, device="cuda")
        rm, rv = torch.zeros(C_, device="cuda"), torch.ones(C_, device="cuda")
        nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
        z = torch.empty(u["N"], C_, *u["dims"], device="cuda")
        count = u["N"] * u["dims"][0] * u["dims"][1] * u["dims"][2]
        outs.append((z, small, rm, rv, nbt))
        calls.append(dict(stats=u["stats"], C=C_, ntiles=u["ntiles"], count=count,
                          bn=(u["gamma"], u["beta"], rm, rv, nbt, 0.1, 1e-5),
                          small=(small[0], small[1], small[2], small[3]), y=u["y"], z=z, relu=True,
                          c0=u["c0"], c_total=u["c_total"]))
    if multi:
        ops.bn_finalize_apply_multi(calls)
    else:
        for c in calls:
            gamma, beta, rm, rv, nbt, mom, eps = c["bn"]
            ops.bn_finalize_apply(c["stats"], c["C"], c["ntiles"], c["count"], gamma, beta, rm, rv, nbt, mom,
                                  eps, *c["small"], c["y"], c["z"], True, c0=c["c0"], c_total=c["c_total"])
    torch.cuda.synchronize()
    return outs


BN_CASES = [
    ("heads 4b", [(B, 192, (8, 8, 8)), (B, 96, (8, 8, 8)), (B, 16, (8, 8, 8))], True),
    ("pair 4b", [(B, 208, (8, 8, 8)), (B, 48, (8, 8, 8))], False),
    ("pair 5c", [(B, 384, (4, 4, 4)), (B, 128, (4, 4, 4))], False),
    ("five units", [(B, 24, (4, 4, 4))] * 5, False),
    # 16x16x16 maps are beyond the one-workgroup-per-channel form: single-unit launches inside the call
    ("large + small", [(B, 32, (16, 16, 16)), (B, 48, (8, 8, 8)), (B, 64, (8, 8, 8))], False),
    ("odd plane (scalar path)", [(4, 8, (1, 3, 3)), (4, 5, (1, 3, 3))], False),
]


@pytest.mark.parametrize("name,shapes,slices", BN_CASES, ids=[c[0] for c in BN_CASES])
def test_batchnorm_multi_is_bit_identical(name, shapes, slices):
    from coclr_amd import ops
    units = _bn_units(5, shapes, slices)
    one = _bn_forward(units, multi=False)
    many = _bn_forward(units, multi=True)
    for a, b in zip(one, many):
        for t1, t2 in zip(a, b):
            assert torch.equal(t1, t2), name
    # backward: dy, dgamma, dbeta
    g = torch.Generator(device="cuda").manual_seed(9)
    res = []
    for multi in (False, True):
        calls, outs = [], []
        for u, (z, small, rm, rv, nbt) in zip(units, one):
            dz = torch.randn(z.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
            dy = torch.empty_like(z)
            dgb = torch.empty(2, u["C"], device="cuda")
            sums = torch.empty(ops.bn_backward_workspace(u["N"], u["C"]), dtype=torch.float64, device="cuda")
            calls.append(dict(dz=dz, y=u["y"], scale=small[2], shift=small[3], mean=small[0], invstd=small[1],
                              sums=sums, dy=dy, dgamma=dgb[0], dbeta=dgb[1], relu=True, training=True))
            outs.append((dy, dgb))
        if multi:
            ops.bn_act_backward_multi(calls)
        else:
            for c in calls:
                ops.bn_act_backward(c["dz"], c["y"], None, c["scale"], c["shift"], c["mean"], c["invstd"],
                                    c["sums"], c["dy"], None, c["dgamma"], c["dbeta"], True, True)
        torch.cuda.synchronize()
        res.append(outs)
    for (dy1, dgb1), (dy2, dgb2) in zip(*res):
        assert torch.equal(dy1, dy2) and torch.equal(dgb1, dgb2), name


@pytest.mark.parametrize("block,dims", [("Mixed_4b", (8, 8, 8)), ("Mixed_4f", (8, 8, 8)), ("Mixed_5c", (4, 4, 4)),
                                        ("Mixed_3b", (16, 16, 16))])
def test_paired_inception_block_matches_unpaired(block, dims, monkeypatch):
    """One inception block, forward and backward, with the branch tails emitted in lockstep (default) and
    one unit after the other: output, input gradient, every parameter gradient and every BatchNorm buffer
    bit-identical; so is the library-level switch COCLR_PAIR=0."""
    from coclr_amd import engine
    from coclr_amd.backbone import s3dg
    from test_gpu_engine import randomise
    cin, widths = s3dg._INCEPTION[block]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, cin, *dims, generator=g).relu()
    dout = None
    results = []
    for pair_units, lib_pair in ((True, "1"), (False, "1"), (True, "0")):
        monkeypatch.setattr(engine, "PAIR_UNITS", pair_units)
        monkeypatch.setenv("COCLR_PAIR", lib_pair)
        torch.manual_seed(0)
        m = s3dg.SepInception(cin, list(widths))
        randomise(m, 7)
        m = m.cuda().train()
        xg = x.cuda().requires_grad_(True)
        out = m(xg)
        if dout is None:
            dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(4)).cuda()
        out.backward(dout)
        torch.cuda.synchronize()
        results.append((out.detach().clone(), xg.grad.clone(),
                        {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: v.clone() for k, v in m.named_buffers()}))
    ref = results[0]
    for other in results[1:]:
        assert torch.equal(ref[0], other[0]) and torch.equal(ref[1], other[1])
        for k in ref[2]:
            assert torch.equal(ref[2][k], other[2][k]), k
        for k in ref[3]:
            assert torch.equal(ref[3][k], other[3][k]), k


# ---- BatchNorm backward sums formed by the data gradient that writes dz (ABI 16) -------------------------

def _bwd_bn_operands(shape, seed, relu=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    C_ = shape[1]
    by = torch.randn(*shape, device="cuda", generator=g)
    scale = torch.randn(C_, device="cuda", generator=g)
    shift = torch.randn(C_, device="cuda", generator=g) * 0.3
    mean = torch.randn(C_, device="cuda", generator=g) * 0.2
    invstd = torch.rand(C_, device="cuda", generator=g) + 0.5
    return (by, scale, shift, mean, invstd, relu)


def _reference_sums(dz, bb):
    by, scale, shift, mean, invstd, relu = bb
    b = lambda v: v.view(1, -1, 1, 1, 1)
    g = dz.double()
    if relu:
        g = g * (torch.addcmul(b(shift), by, b(scale)) > 0)     # the kernels' fmaf
    xhat = ((by - b(mean)) * b(invstd)).double()
    return g.sum((0, 2, 3, 4)), (g * xhat).sum((0, 2, 3, 4))


SUMS = [
    # the temporal half of a separable unit: F(2,3) data gradient -> the unit's bn1 (Conv_2c, Mixed_3b/3c sizes)
    ("2c conv2", (192, 192, (3, 1, 1), (1, 0, 0), (16, 32, 32))),
    ("3c b1 conv2", (192, 192, (3, 1, 1), (1, 0, 0), (16, 16, 16))),
    ("3b b2 conv2", (32, 32, (3, 1, 1), (1, 0, 0), (16, 16, 16))),
    ("odd frames", (48, 40, (3, 1, 1), (1, 0, 0), (7, 8, 8))),
    # direct kernels: small-map (1,3,3) and pointwise
    ("4b b1 conv1", (96, 208, (1, 3, 3), (0, 1, 1), (8, 8, 8))),
    ("4b heads", (480, 304, (1, 1, 1), (0, 0, 0), (8, 8, 8))),
]


@pytest.mark.parametrize("name,spec", SUMS, ids=[s[0] for s in SUMS])
@pytest.mark.parametrize("relu", [True, False])
def test_data_gradient_forms_batchnorm_backward_sums(name, spec, relu):
    """coclr_conv_call.bwd_y: dz is bit-identical to the plain data gradient, the folded partial sums are
    the float64 sums over the whole tensor (1e-5 of the sum of magnitudes: fp32 partials per tile), and
    coclr_bn_act_backward_multi on those partials gives the dy / dgamma / dbeta of its own reduction."""
    from coclr_amd import engine, ops
    run = engine.Run(torch.device("cuda"), save=False)
    case = _conv_case(run, *spec, seed=31, dgrad=True)
    geom = case["geom"]
    if geom.algo == 2:
        # the F(4,3) kernel (16-frame temporal layers) does not form the sums -- the engine asks
        # (geom.bwd_sums_ok()) and keeps the reduction pass there; the F(2,3) form of the same layer does
        assert not geom.bwd_sums_ok()
        case = _conv_case(run, *spec, seed=31, dgrad=True, algo=1)
        geom = case["geom"]
    assert geom.bwd_sums_ok()
    plain, _ = _run_single(case)
    bb = _bwd_bn_operands(tuple(plain.shape), 32, relu)
    dz = case["y0"].clone()
    nt = geom.ntiles()
    st = torch.full((2 * geom.Cout * nt,), 7.0, device="cuda")
    ops.conv_fwd_multi([dict(geom=geom, x=case["x"], w=case["w"], y=dz, stats=st, bwd_bn=bb)])
    torch.cuda.synchronize()
    assert torch.equal(dz, plain)
    sg, sgx = _reference_sums(dz, bb)
    got = st.view(2, geom.Cout, nt).double().sum(2)
    mag_g, mag_gx = _reference_sums(dz.abs(), (bb[0], bb[1], bb[2], bb[3], bb[4], relu))
    assert ((got[0] - sg).abs() <= 1e-5 * mag_g + 1e-6).all()
    by, scale, shift, mean, invstd, _ = bb
    xabs = ((by - mean.view(1, -1, 1, 1, 1)) * invstd.view(1, -1, 1, 1, 1)).abs().double()
    gabs = dz.abs().double()
    assert ((got[1] - sgx).abs() <= 1e-5 * (gabs * xabs).sum((0, 2, 3, 4)) + 1e-6).all()

    # the unit's backward from those partials against its own reduction pass
    N, C_ = dz.shape[0], dz.shape[1]
    outs = []
    for parts in (None, [(st, nt)]):
        dy = torch.empty_like(dz)
        dgamma, dbeta = torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
        sums = torch.empty(ops.bn_backward_workspace(N, C_), dtype=torch.float64, device="cuda")
        ops.bn_act_backward_multi([dict(dz=dz, y=by, scale=scale, shift=shift, mean=mean, invstd=invstd,
                                        sums=sums, dy=dy, dgamma=dgamma, dbeta=dbeta, relu=relu, training=True,
                                        partials=parts)])
        outs.append((dy, dgamma, dbeta))
    torch.cuda.synchronize()
    for a, b_ in zip(*outs):
        assert (a - b_).abs().max() <= 2e-5 * b_.abs().max() + 1e-6


def test_strided_phases_form_the_sums_together():
    """Conv_1a.conv2 is (7,1,1) stride 2: its data gradient is one launch per residue class of dX
    (ConvGeom.dgrad_phases), each with its own partial sums; the fold takes both."""
    from coclr_amd import engine, ops
    run = engine.Run(torch.device("cuda"), save=False)
    g = torch.Generator(device="cuda").manual_seed(3)
    N, Cc, idim = 8, 64, (16, 32, 32)
    geom = ops.conv_geom(N, Cc, Cc, idim, (7, 1, 1), (2, 1, 1), (3, 0, 0))
    w = torch.randn(Cc, Cc, 7, 1, 1, device="cuda", generator=g) * 0.05
    dyo = torch.randn(N, Cc, *geom.odim, device="cuda", generator=g)
    phases = geom.dgrad_phases()
    assert phases is not None and len(phases) == 2 and all(ph[0].bwd_sums_ok() for ph in phases)
    plain = torch.zeros(N, Cc, *idim, device="cuda")
    dz = torch.zeros_like(plain)
    bb = _bwd_bn_operands(tuple(dz.shape), 4)
    parts = []
    for pg, k0, nk, step in phases:
        wp = run.pack(w, True, taps=nk, tap_base=k0, tap_step=step)
        ops.conv_fwd(pg, dyo, wp, plain)
        nt = pg.ntiles()
        st = torch.empty(2 * Cc * nt, device="cuda")
        ops.conv_fwd_multi([dict(geom=pg, x=dyo, w=wp, y=dz, stats=st, bwd_bn=bb)])
        parts.append((st, nt))
    torch.cuda.synchronize()
    assert torch.equal(dz, plain)
    sg, sgx = _reference_sums(dz, bb)
    got = sum(st.view(2, Cc, nt).double().sum(2) for st, nt in parts)
    assert ((got[0] - sg).abs() <= 1e-5 * dz.abs().double().sum((0, 2, 3, 4))).all()
    outs = []
    for pp in (None, parts):
        dy = torch.empty_like(dz)
        dgamma, dbeta = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
        sums = torch.empty(ops.bn_backward_workspace(N, Cc), dtype=torch.float64, device="cuda")
        ops.bn_act_backward_multi([dict(dz=dz, y=bb[0], scale=bb[1], shift=bb[2], mean=bb[3], invstd=bb[4],
                                        sums=sums, dy=dy, dgamma=dgamma, dbeta=dbeta, relu=True, training=True,
                                        partials=pp)])
        outs.append((dy, dgamma, dbeta))
    torch.cuda.synchronize()
    for a, b_ in zip(*outs):
        assert (a - b_).abs().max() <= 2e-5 * b_.abs().max() + 1e-6


def test_backward_sums_in_pair_launches_and_refusals():
    """Two temporal data gradients in one launch, each with its own unit's sums; the spatial Winograd kernel
    has no such epilogue and says so (bwd_sums_ok False, COCLR_EINVAL when asked anyway), and the
    combination with an accumulating destination is refused."""
    from coclr_amd import engine, ops, _lib
    run = engine.Run(torch.device("cuda"), save=False)
    cases = [_conv_case(run, *spec, seed=41 + i, dgrad=True)
             for i, spec in enumerate([(208, 208, (3, 1, 1), (1, 0, 0), (8, 8, 8)),
                                       (48, 48, (3, 1, 1), (1, 0, 0), (8, 8, 8))])]
    singles, pairs = [], []
    for c in cases:
        bb = _bwd_bn_operands(tuple(c["y0"].shape), 50 + c["geom"].Cout)
        c["bb"] = bb
        nt = c["geom"].ntiles()
        dz, st = c["y0"].clone(), torch.empty(2 * c["geom"].Cout * nt, device="cuda")
        ops.conv_fwd_multi([dict(geom=c["geom"], x=c["x"], w=c["w"], y=dz, stats=st, bwd_bn=bb)])
        singles.append((dz, st))
    ys = [c["y0"].clone() for c in cases]
    sts = [torch.empty_like(s[1]) for s in singles]
    ops.conv_fwd_multi([dict(geom=c["geom"], x=c["x"], w=c["w"], y=y, stats=st, bwd_bn=c["bb"])
                        for c, y, st in zip(cases, ys, sts)])
    torch.cuda.synchronize()
    for (dz, st), y, s2 in zip(singles, ys, sts):
        assert torch.equal(dz, y) and torch.equal(st, s2)

    wide = _conv_case(run, 64, 192, (1, 3, 3), (0, 1, 1), (16, 32, 32), seed=60, dgrad=True)
    if wide["geom"].algo == 1:
        assert not wide["geom"].bwd_sums_ok()
        bb = _bwd_bn_operands(tuple(wide["y0"].shape), 61)
        st = torch.empty(2 * wide["geom"].Cout * wide["geom"].ntiles(), device="cuda")
        with pytest.raises(_lib.HipLibraryError):
            ops.conv_fwd_multi([dict(geom=wide["geom"], x=wide["x"], w=wide["w"], y=wide["y0"].clone(),
                                     stats=st, bwd_bn=bb)])
    c = cases[0]
    with pytest.raises(_lib.HipLibraryError):
        ops.conv_fwd_multi([dict(geom=c["geom"], x=c["x"], w=c["w"], y=c["y0"].clone(), stats=sts[0],
                                 bwd_bn=c["bb"], accumulate=True)])


def test_engine_with_the_sums_epilogue_matches_the_reduction_pass(monkeypatch):
    """COCLR_FUSE_BN_REDUCE=1 (opt-in): the S3D backbone's backward with Conv_1a.bn1 / Conv_2c.bn1 taking their
    sums from the data gradients' epilogues against the default reduction passes -- same forward, same ReLU
    masks, sums folded from fp32 tile partials instead of per-sample fp64 partials: every parameter gradient
    within 1e-5 of the tensor's largest entry."""
    from coclr_amd import engine, ops
    from coclr_amd.backbone import s3dg
    used = {"n": 0}
    real = ops.bn_act_backward_multi

    def spy(units):
        used["n"] += sum(1 for u in units if u.get("partials"))
        return real(units)

    monkeypatch.setattr(ops, "bn_act_backward_multi", spy)
    # Conv_2c.conv2's data gradient in its F(2,3) form: the F(4,3) kernel the 16-frame layers take by default does
    # not form the sums (the engine then keeps the reduction pass for Conv_2c.bn1)
    monkeypatch.setattr(ops, "WINOGRAD_T4", False)
    ops._GEOMS.clear()
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(16, 3, 32, 64, 64, device="cuda", generator=g)     # Conv_2c: 65536 values per channel
    grads = []
    for fuse in (False, True):
        monkeypatch.setattr(engine, "FUSE_BN_REDUCE", fuse)
        torch.manual_seed(0)
        m = s3dg.S3D(input_channel=3).cuda().train()
        used["n"] = 0
        out = m(x)
        out.square().mean().backward()
        torch.cuda.synchronize()
        grads.append(({k: p.grad.clone() for k, p in m.named_parameters()}, out.detach().clone(), used["n"]))
    (ga, oa, na), (gb, ob, nb) = grads
    assert na == 0 and nb >= 2, (na, nb)
    assert torch.equal(oa, ob)
    for k in ga:
        assert (ga[k] - gb[k]).abs().max() <= 1e-5 * ga[k].abs().max() + 1e-12, k
    ops._GEOMS.clear()        # geometries cached with the F(2,3) choice must not leak into later tests
