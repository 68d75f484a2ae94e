"""Parity at BASELINE.json's FULL sizes (B=32 clips of 3x32x128x128, K=16384) through
size-independent properties -- the CPU oracle would need ~20 s per step there:

  * every convolution geometry of the S3D backbone: the adjoint identities
        <conv(x, w), dy> == <x, dgrad(dy, w)> == <w, wgrad(x, dy)>
    tie the three kernels (forward / data gradient incl. the phase-decomposed strided form /
    weight gradient incl. split-K) to each other; the forward itself is pinned to the oracle at
    small sizes (test_gpu_kernels.py) and to the reference at B=4 full resolution
    (test_gpu_model.py::test_config1_matches_reference);
  * BatchNorm statistics from the conv epilogue == mean / biased variance of the output;
  * max-pool: y == x[argmax], backward routes every dy to exactly one input (sum preserved);
  * queue: K/B enqueues of a K=16384 queue are a FIFO round trip (bit-exact), pointer wraps.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B = 32


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _s3d_geometries():
    """Unique (conv geometry, is-first-layer) pairs of one S3D forward at the benchmark shape,
    recorded from the engine itself."""
    from coclr_amd import ops
    from backbone.select_backbone import select_backbone
    seen, order = set(), []
    inner = ops.conv_fwd

    def rec(geom, *a, **kw):
        key = (geom.Cin, geom.Cout, geom.idim, geom.k, geom.s, geom.p)
        if key not in seen:
            seen.add(key)
            order.append(key)
        return inner(geom, *a, **kw)

    ops.conv_fwd = rec
    try:
        torch.manual_seed(0)
        net, _ = select_backbone("s3d")
        net = net.cuda().train()
        with torch.no_grad():
            net(torch.randn(B, 3, 32, 128, 128, device="cuda"))
    finally:
        ops.conv_fwd = inner
    return order


def test_conv_adjoint_identities_every_s3d_layer_full_size():
    from coclr_amd import ops, engine
    run = engine.Run(torch.device("cuda"), save=False)
    geoms = _s3d_geometries()
    assert len(geoms) >= 50
    g0 = torch.Generator(device="cuda").manual_seed(7)
    worst = 0.0
    for (cin, cout, idim, k, s, p) in geoms:
        g = ops.ConvGeom(B, cin, cout, idim, k, s, p)
        x = torch.randn(B, cin, *idim, device="cuda", generator=g0)
        w = torch.randn(cout, cin, *k, device="cuda", generator=g0) * 0.05
        dy = torch.randn(B, cout, *g.odim, device="cuda", generator=g0)
        y = torch.empty_like(dy)
        ops.conv_fwd(g, x, run.pack(w, False), y)
        ref = _dot(y, dy)
        scale = float(y.double().norm() * dy.double().norm()) + 1e-30
        # data gradient (phase-decomposed when the engine would use that form)
        dx = torch.full_like(x, float("nan"))
        phases = g.dgrad_phases()
        if phases is not None:
            for pg, k0, nk, step in phases:
                ops.conv_fwd(pg, dy, run.pack(w, True, taps=nk, tap_base=k0, tap_step=step), dx)
        else:
            ops.conv_fwd(g.dgrad(), dy, run.pack(w, True), dx)
        e_d = abs(_dot(x, dx) - ref) / scale
        # weight gradient
        dw = torch.empty_like(w)
        ws = torch.empty(g.wgrad_workspace(), device="cuda")
        kk = k[0] * k[1] * k[2]
        ops.conv_wgrad(g, x, dy, dw, ws, cin * kk, kk, 0)
        e_w = abs(_dot(w, dw) - ref) / scale
        worst = max(worst, e_d, e_w)
        # fp32 products summed over up to 1e9 terms: 1e-5 of the Cauchy-Schwarz scale
        assert e_d <= 1e-5 and e_w <= 1e-5, (cin, cout, idim, k, s, e_d, e_w)
        del x, w, dy, y, dx, dw, ws
    print("worst adjoint mismatch (relative to |y||dy|): %.2e over %d geometries" % (worst, len(geoms)))


def test_conv_epilogue_statistics_full_size():
    """stats of the stem conv (persistent kernel, one partial per workgroup) and of Conv_2c.conv1."""
    from coclr_amd import ops, engine
    run = engine.Run(torch.device("cuda"), save=False)
    g0 = torch.Generator(device="cuda").manual_seed(8)
    for (cin, cout, idim, k, s, p) in [(3, 64, (32, 128, 128), (1, 7, 7), (1, 2, 2), (0, 3, 3)),
                                        (64, 192, (16, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1))]:
        g = ops.ConvGeom(B, cin, cout, idim, k, s, p)
        x = torch.randn(B, cin, *idim, device="cuda", generator=g0) + 0.3
        w = torch.randn(cout, cin, *k, device="cuda", generator=g0) * 0.05
        y = torch.empty(B, cout, *g.odim, device="cuda")
        st = torch.empty(2 * cout * g.ntiles(), device="cuda")
        ops.conv_fwd(g, x, run.pack(w, False), y, stats=st)
        cnt = y.numel() // cout
        small = torch.empty(4, cout, device="cuda")
        rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda")
        ops.bn_finalize(st, cout, g.ntiles(), cnt, torch.ones(cout, device="cuda"),
                        torch.zeros(cout, device="cuda"), rm, rv, nbt, 0.1, 1e-5, small[0], small[1],
                        small[2], small[3])
        yd = y.double()
        mean = yd.mean((0, 2, 3, 4))
        var = yd.var((0, 2, 3, 4), unbiased=False)
        assert float((small[0].double() - mean).abs().max() / mean.abs().max()) < 1e-5
        assert float(((1.0 / small[1].double() ** 2 - 1e-5) / var - 1).abs().max()) < 1e-4


def test_maxpool_properties_full_size():
    from coclr_amd import ops
    g0 = torch.Generator(device="cuda").manual_seed(9)
    for (c, idim, k, s, p) in [(64, (16, 64, 64), (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                               (192, (16, 16, 16), (3, 3, 3), (1, 1, 1), (1, 1, 1)),
                               (480, (16, 16, 16), (3, 3, 3), (2, 2, 2), (1, 1, 1))]:
        g = ops.PoolGeom(B, c, idim, k, s, p)
        x = torch.randn(B, c, *idim, device="cuda", generator=g0)
        y = torch.empty(B, c, *g.odim, device="cuda")
        idx = torch.empty(B, c, *g.odim, dtype=torch.int32, device="cuda")
        ops.maxpool_fwd(g, x, y, idx)
        picked = x.flatten(2).gather(2, idx.flatten(2).long()).view_as(y)
        assert torch.equal(picked, y)                      # y is x at the recorded argmax
        assert float(y.min()) >= float(x.min())
        dy = torch.randn_like(y)
        dx = torch.empty_like(x)
        ops.maxpool_bwd(g, dy, idx, dx)
        assert abs(float(dx.double().sum()) - float(dy.double().sum())) <= 1e-6 * float(dy.double().abs().sum())
        assert int((dx != 0).sum()) <= dy.numel()


@pytest.mark.parametrize("K,world", [(2048, 1), (16384, 8)])
def test_queue_fifo_round_trip(K, world):
    """dequeue/enqueue (model/pretrain.py:82-96) at the benchmark queue sizes: K/(B*world)
    enqueues of random unit keys overwrite the whole queue in order and wrap the pointer."""
    from coclr_amd import ops
    D, BW = 128, B * world
    g0 = torch.Generator(device="cuda").manual_seed(10)
    queue = torch.randn(D, K, device="cuda", generator=g0)
    ptr = torch.zeros(1, dtype=torch.int64, device="cuda")
    label = torch.full((K,), -1, dtype=torch.int64, device="cuda")
    allk, alll = [], []
    for i in range(K // BW):
        keys = torch.nn.functional.normalize(torch.randn(BW, D, device="cuda", generator=g0), dim=1)
        lab = torch.randint(0, 1000, (BW,), device="cuda", generator=g0)
        ops.queue_enqueue(queue, keys, ptr)
        ops.queue_fill_i64(label, lab, 0, BW, ptr)
        ops.queue_advance(ptr, BW, K)
        assert int(ptr) == ((i + 1) * BW) % K
        allk.append(keys)
        alll.append(lab)
    assert torch.equal(queue, torch.cat(allk, 0).t())       # bit exact: pure data movement
    assert torch.equal(label, torch.cat(alll, 0))
    assert int(ptr) == 0
