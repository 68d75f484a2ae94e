"""Parity at BASELINE.json's FULL sizes (B=32 clips of 3x32x128x128, K=16384) through
size-independent properties -- the CPU oracle would need ~20 s per step there:

  * every convolution geometry of the S3D and ResNet2d3d-50 backbones: the adjoint identities
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


def _geometries(network):
    """Unique conv geometries of one backbone forward at the benchmark shape, recorded from the
    engine itself (so strided 1x1x1 downsamples appear as the dense pointwise convs they run as and
    r50's (5,7,7) stem as its five per-tap (1,7,7) launches with their shifted padding)."""
    from coclr_amd import ops
    from backbone.select_backbone import select_backbone
    seen, order = set(), []
    inner = ops.conv_fwd

    def rec(geom, *a, **kw):
        key = (geom.Cin, geom.Cout, geom.idim, geom.k, geom.s, geom.p, geom.odim)
        if key not in seen:
            seen.add(key)
            order.append(key)
        return inner(geom, *a, **kw)

    inner_multi = ops.conv_fwd_multi

    def rec_multi(calls):
        # sibling units emitted in lockstep come through the multi entry point
        for c in calls:
            geom = c["geom"]
            key = (geom.Cin, geom.Cout, geom.idim, geom.k, geom.s, geom.p, geom.odim)
            if key not in seen:
                seen.add(key)
                order.append(key)
        return inner_multi(calls)

    ops.conv_fwd, ops.conv_fwd_multi = rec, rec_multi
    try:
        torch.manual_seed(0)
        net, _ = select_backbone(network)
        net = net.cuda().train()
        with torch.no_grad():
            net(torch.randn(B, 3, 32, 128, 128, device="cuda"))
    finally:
        ops.conv_fwd, ops.conv_fwd_multi = inner, inner_multi
    del net
    torch.cuda.empty_cache()
    return order


@pytest.mark.parametrize("network,at_least", [("s3d", 50), ("r50", 20)])
def test_conv_adjoint_identities_every_layer_full_size(network, at_least):
    """S3D (BASELINE configs 2-4) and ResNet2d3d-50 (config 5: Cin/Cout up to 2048, the (1,3,3)/2
    zero-upsampled data gradient, the five-launch (5,7,7) stem) at B=32."""
    from coclr_amd import ops, engine
    run = engine.Run(torch.device("cuda"), save=False)
    geoms = _geometries(network)
    assert len(geoms) >= at_least
    g0 = torch.Generator(device="cuda").manual_seed(7)
    worst = 0.0
    for (cin, cout, idim, k, s, p, odim) in geoms:
        g = ops.ConvGeom(B, cin, cout, idim, k, s, p, odim=odim)
        x = torch.randn(B, cin, *idim, device="cuda", generator=g0)
        w = torch.randn(cout, cin, *k, device="cuda", generator=g0) * 0.05
        dy = torch.randn(B, cout, *g.odim, device="cuda", generator=g0)
        y = torch.empty_like(dy)
        ops.conv_fwd(g, x, run.pack(w, False), y)
        ref = _dot(y, dy)
        scale = float(y.double().norm() * dy.double().norm()) + 1e-30
        # data gradient (phase-decomposed when the engine would use that form); the network input
        # never needs one (and the temporal slices of r50's stem have negative padding)
        e_d = 0.0
        dx = None
        if cin != 3:
            dx = torch.full_like(x, float("nan"))
            phases = g.dgrad_phases()
            if phases is not None:
                for pg, k0, nk, step in phases:
                    ops.conv_fwd(pg, dy, run.pack(w, True, taps=nk, tap_base=k0, tap_step=step, algo=pg.algo), dx)
            else:
                ops.conv_fwd(g.dgrad(), dy, run.pack(w, True), dx)
            e_d = abs(_dot(x, dx) - ref) / scale
        # weight gradient (direct form, and the form the engine's geometry selects when it differs:
        # Winograd F(2,3) for the wide (3,1,1) layers)
        dw = torch.empty_like(w)
        ws = torch.empty(g.wgrad_workspace(), device="cuda")
        kk = k[0] * k[1] * k[2]
        ops.conv_wgrad(g, x, dy, dw, ws, cin * kk, kk, 0)
        e_w = abs(_dot(w, dw) - ref) / scale
        ge = ops.conv_geom(B, cin, cout, idim, k, s, p) if min(p) >= 0 else g
        if ge.algo != g.algo:
            # the forward and data gradient in the selected (Winograd) form too
            y2 = torch.empty_like(y)
            ops.conv_fwd(ge, x, run.pack(w, False, algo=ge.algo), y2)
            e_w = max(e_w, float((y2 - y).abs().max() / y.abs().max()) * 1e-2)         # elementwise 1e-3
            dg = ge.dgrad()
            if dg.algo:
                dx2 = torch.empty_like(x)
                ops.conv_fwd(dg, dy, run.pack(w, True, algo=dg.algo), dx2)
                e_d = max(e_d, abs(_dot(x, dx2) - ref) / scale)
                del dx2
            dw2 = torch.empty_like(w)
            ws2 = torch.empty(ge.wgrad_workspace(), device="cuda")
            ops.conv_wgrad(ge, x, dy, dw2, ws2, cin * kk, kk, 0)
            e_w = max(e_w, abs(_dot(w, dw2) - ref) / scale)
            e_w = max(e_w, float((dw2 - dw).abs().max() / dw.abs().max()) * 1e-2)   # elementwise 1e-3
            del dw2, ws2, y2
        worst = max(worst, e_d, e_w)
        # fp32 products summed over up to 1e9 terms: 1e-5 of the Cauchy-Schwarz scale
        assert e_d <= 1e-5 and e_w <= 1e-5, (cin, cout, idim, k, s, p, e_d, e_w)
        del x, w, dy, y, dx, dw, ws
    print("%s: worst adjoint mismatch (relative to |y||dy|): %.2e over %d geometries"
          % (network, worst, len(geoms)))


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


# ---- the contrastive head at the queue sizes of BASELINE configs 2-5 ---------------------------

@pytest.mark.parametrize("K", [2048, 16384])
def test_nce_logits_forward_backward_at_config_sizes(K):
    """[l_pos | q.queue]/T (model/pretrain.py:175-182) and its backward at B=32, dim=128 against
    the einsum formulation evaluated on the host."""
    import torch.nn.functional as F
    from coclr_amd import ops
    g = torch.Generator().manual_seed(20 + K)
    D, T = 128, 0.07
    x = torch.randn(B, D, generator=g).requires_grad_(True)
    q = F.normalize(x, dim=1)
    k = F.normalize(torch.randn(B, D, generator=g), dim=1)
    queue = F.normalize(torch.randn(D, K, generator=g), dim=0)
    logits = torch.cat([torch.einsum('nc,nc->n', [q, k]).unsqueeze(-1),
                        torch.einsum('nc,ck->nk', [q, queue])], 1) / T
    dl = torch.randn(B, 1 + K, generator=g) / K
    logits.backward(dl)
    xd = x.detach().cuda()
    qd, inv = torch.empty_like(xd), torch.empty(B, device="cuda")
    ops.l2norm_fwd(xd, qd, inv)
    lg = torch.empty(B, 1 + K, device="cuda")
    ops.nce_logits_fwd(qd, k.cuda(), queue.cuda(), lg, T)
    err = float((lg.cpu() - logits.detach()).abs().max() / logits.detach().abs().max())
    assert err <= 2e-5, err
    dq = torch.empty(B, D, device="cuda")
    splits = max(1, min(K // 128, 256))
    ws = torch.empty(max(1, ops.gemm_workspace(B, D, K, splits)), device="cuda")
    ops.nce_logits_bwd(dl.cuda(), k.cuda(), queue.cuda(), dq, ws, T, splits)
    dx = torch.empty_like(xd)
    ops.l2norm_bwd(dq, qd, inv, dx)
    err = float((dx.cpu() - x.grad).abs().max() / x.grad.abs().max())
    assert err <= 2e-4, err


@pytest.mark.parametrize("K", [2048, 16384])
def test_cross_modal_mining_at_config_sizes(K):
    """CoCLR positive mining (model/pretrain.py:405-413) at B=32: similarity GEMM against the
    second queue, same-source entries masked with -inf, top-5 per row OR-ed into the mask -- exact
    against torch.topk / scatter_, including a row with fewer than 5 finite candidates and a row
    with none (K=16384 is the 64 KiB dynamic-LDS row of the mining kernel)."""
    import torch.nn.functional as F
    from coclr_amd import ops
    g = torch.Generator().manual_seed(30 + K)
    D, topk = 128, 5
    kf = F.normalize(torch.randn(B, D, generator=g), dim=1)
    queue2 = F.normalize(torch.randn(D, K, generator=g), dim=0)
    names = torch.randint(0, 400, (K,), generator=g)
    src = torch.randint(0, 400, (B,), generator=g)
    names[:7] = -1                                     # never matches (queue_vname init, ref :312)
    row1_free = torch.tensor([5, K // 2, K - 1])
    sim_ref = kf @ queue2
    sim = torch.empty(B, K, device="cuda")
    ops.gemm(kf.cuda(), D, 1, queue2.cuda(), K, 1, sim, K, None, B, K, D)
    err = float((sim.cpu() - sim_ref).abs().max() / sim_ref.abs().max())
    assert err <= 2e-5, err
    mask = torch.empty(B, 1 + K, dtype=torch.uint8, device="cuda")
    ops.positive_mask(sim, src.cuda(), names.cuda(), mask, topk)
    got = mask.cpu().bool()
    # reference semantics on the kernel's own similarity values (ties aside, fp32 reassociation
    # could otherwise reorder near-equal candidates)
    simk = sim.cpu()
    same_k = src[:, None] == names[None, :]
    ms = simk.clone()
    ms[same_k] = -float("inf")
    _, idx = torch.topk(ms, topk, dim=1)
    exp = same_k.clone()
    exp.scatter_(1, idx, True)
    exp = torch.cat([torch.ones(B, 1, dtype=torch.bool), exp], 1)
    assert torch.equal(got, exp)
    # the fused form (coclr_mine_positives: similarity tiles + running top-k, one launch, no sim
    # tensor): same mask; its similarities (optional debug output) equal the GEMM's to round-off
    ws = ops.mine_workspace(B, K, topk, "cuda")
    sim2 = torch.empty(B, K, device="cuda")
    for rep in range(2):                               # twice: the counters must come back to zero
        mask2 = torch.full((B, 1 + K), 7, dtype=torch.uint8, device="cuda")
        ops.mine_positives(kf.cuda(), queue2.cuda(), src.cuda(), names.cuda(), mask2, topk, ws, sim_out=sim2)
        s2 = sim2.cpu()
        assert float((s2 - sim_ref).abs().max() / sim_ref.abs().max()) <= 2e-5
        ms2 = s2.clone()
        ms2[same_k] = -float("inf")
        _, idx2 = torch.topk(ms2, topk, dim=1)
        exp2 = same_k.clone()
        exp2.scatter_(1, idx2, True)
        exp2 = torch.cat([torch.ones(B, 1, dtype=torch.bool), exp2], 1)
        assert torch.equal(mask2.cpu().bool(), exp2), "fused mining, pass %d" % rep
        assert int(ws[2].abs().sum()) == 0
    # fused form on rows with fewer than topk (3) and no (0) non-sibling columns: siblings only + the 3
    for nfree in (3, 0):
        nm = src[:1].repeat(K)
        free = torch.tensor([5, K // 2, K - 1])[:nfree]
        nm[free] = -5
        m1 = torch.empty(1, 1 + K, dtype=torch.uint8, device="cuda")
        ops.mine_positives(kf[:1].cuda(), queue2.cuda(), src[:1].cuda(), nm.cuda(), m1, topk,
                           ops.mine_workspace(1, K, topk, "cuda"))
        assert int(m1.sum()) == 1 + K, "every column is a sibling or one of the %d free ones" % nfree
    # rows with fewer than topk finite candidates, through single-row launches with their own
    # name tables: 3 free columns -> exactly those 3 plus two -inf picks (torch.topk takes the
    # lowest-index -inf entries; so does the kernel), 0 free columns -> the siblings only plus
    # topk -inf picks
    for row, free in ((1, row1_free), (2, torch.tensor([], dtype=torch.long))):
        nm = torch.full((K,), int(src[row]), dtype=torch.long)
        nm[free] = -5
        m1 = torch.empty(1, 1 + K, dtype=torch.uint8, device="cuda")
        ops.positive_mask(sim[row:row + 1].contiguous(), src[row:row + 1].cuda(), nm.cuda(), m1, topk)
        same1 = (nm == src[row])[None, :]
        ms1 = simk[row:row + 1].clone()
        ms1[same1] = -float("inf")
        _, idx1 = torch.topk(ms1, topk, dim=1)
        exp1 = same1.clone()
        exp1.scatter_(1, idx1, True)
        exp1 = torch.cat([torch.ones(1, 1, dtype=torch.bool), exp1], 1)
        assert torch.equal(m1.cpu().bool(), exp1), "row with %d free columns" % len(free)
        assert int(exp1.sum()) == 1 + K                 # all siblings (+ the picks among them)


def test_config2_training_step_matches_oracle():
    """BASELINE.json configs[1] -- the benchmarked configuration: S3D InfoNCE, K=2048, B=32 clips
    of 3x32x128x128, one training step (query forward, momentum update, shuffle-BN key forward,
    logits, enqueue, backward) on the GPU against the CPU oracle run on this host's cores with the
    same state, inputs and permutation.  Bars: the north star's 1e-3 on logits / loss / queue
    columns / BatchNorm running statistics, exact pointer and labels."""
    import os
    import torch.nn.functional as F
    import model.pretrain as product
    from oracle import coclr_oracle as orc
    from _cases import check_close
    K = 2048
    torch.manual_seed(0)
    model = product.InfoNCE('s3d', 128, K, 0.999, 0.07)
    ref_sd = orc.training_state(model.state_dict())
    model = model.cuda().train()
    torch.manual_seed(1)
    block = torch.randn(B, 2, 3, 32, 128, 128)
    torch.manual_seed(2)
    perm = torch.randperm(B)
    torch.manual_seed(2)
    logits, labels = model(block.cuda())
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()

    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        (ref_logits, ref_labels), = orc.nce_step(ref_sd, "infonce", "s3d", [block], None, 128, K,
                                                 0.999, 0.07, perm)
        ref_loss = F.cross_entropy(ref_logits, ref_labels)
        ref_loss.backward()
    finally:
        torch.set_num_threads(threads)
    check_close(logits, ref_logits, 1e-3, "logits")
    assert torch.equal(labels.cpu(), ref_labels)
    # loss ~ 1e-2 at initialisation (saturated softmax): relative error of the loss = absolute
    # error of the logit gaps; hold it to the 1e-3 logit bar expressed in loss units
    assert abs(float(loss) - float(ref_loss)) <= 1e-3 * max(1.0, float(ref_logits.abs().max())) * 1.0
    sd = model.state_dict()
    assert int(sd["queue_ptr"]) == int(ref_sd["queue_ptr"]) == B
    check_close(sd["queue"][:, :B], ref_sd["queue"][:, :B], 1e-3, "enqueued keys")
    assert torch.equal(sd["queue"][:, B:].cpu(), ref_sd["queue"][:, B:])      # untouched columns
    for k in ("encoder_q.0.Conv_1a.bn1.running_mean", "encoder_q.0.Conv_1a.bn1.running_var",
              "encoder_q.0.Mixed_5c.branch3.1.bn.running_mean",
              "encoder_q.0.Mixed_5c.branch3.1.bn.running_var",
              "encoder_k.0.Conv_1a.bn1.running_mean", "encoder_k.0.Mixed_5c.branch0.0.bn.running_var",
              "encoder_k.4.bias", "encoder_k.0.Conv_2c.conv1.weight"):
        check_close(sd[k], ref_sd[k], 1e-3, k)
    assert int(sd["encoder_q.0.Conv_2b.bn.num_batches_tracked"]) == 1
    # head gradient is well conditioned at initialisation (tests/_cases.py explains why the
    # backbone's are not)
    g = model.encoder_q[4].weight.grad
    rg = ref_sd["encoder_q.4.weight"].grad
    check_close(g, rg, 5e-3, "head weight gradient")


def test_ubernce_training_step_at_config2_size():
    """UberNCE (model/pretrain.py:193-278, the supervised variant main_nce.py:318-324 trains) at BASELINE
    config 2's size -- B=32 clips of 3x32x128x128, K=2048 -- against the CPU oracle: logits at the north star's
    1e-3, the positive mask (col 0 | k_label == queue_label) and the label queue EXACT, pointer exact, enqueued
    keys and the head's weight gradient under the loss of main_nce.py:318-320.  The label queue is pre-filled
    (51 classes, a stretch still -1) so that the mask has real positives."""
    import os
    import model.pretrain as product
    from oracle import coclr_oracle as orc
    from _cases import check_close, loss_fn
    K, ncls = 2048, 51
    torch.manual_seed(0)
    model = product.UberNCE('s3d', 128, K, 0.999, 0.07)
    g = torch.Generator().manual_seed(3)
    model.queue_label.copy_(torch.randint(0, ncls, (K,), generator=g))
    model.queue_label[K - 200:] = -1
    ref_sd = orc.training_state(model.state_dict())
    model = model.cuda().train()
    torch.manual_seed(1)
    block = torch.randn(B, 2, 3, 32, 128, 128)
    k_label = torch.randint(0, ncls, (B,), generator=g)
    torch.manual_seed(2)
    perm = torch.randperm(B)
    torch.manual_seed(2)
    logits, mask = model(block.cuda(), k_label.cuda())
    loss = loss_fn("ubernce", logits, mask)
    loss.backward()
    torch.cuda.synchronize()

    threads = torch.get_num_threads()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    try:
        (ref_logits, ref_mask), = orc.nce_step(ref_sd, "ubernce", "s3d", [block], [k_label], 128, K, 0.999, 0.07,
                                               perm)
        ref_loss = loss_fn("ubernce", ref_logits, ref_mask)
        ref_loss.backward()
    finally:
        torch.set_num_threads(threads)
    check_close(logits, ref_logits, 1e-3, "logits")
    assert mask.dtype == torch.bool and torch.equal(mask.cpu(), ref_mask.bool())
    assert int(mask[:, 1:].sum()) > B, "the pre-filled label queue should give every row positives"
    assert abs(float(loss) - float(ref_loss)) <= 1e-3 * max(1.0, abs(float(ref_loss)))
    sd = model.state_dict()
    assert int(sd["queue_ptr"]) == int(ref_sd["queue_ptr"]) == B
    assert torch.equal(sd["queue_label"].cpu(), ref_sd["queue_label"])
    assert torch.equal(sd["queue_label"][:B].cpu(), k_label)
    check_close(sd["queue"][:, :B], ref_sd["queue"][:, :B], 1e-3, "enqueued keys")
    assert torch.equal(sd["queue"][:, B:].cpu(), ref_sd["queue"][:, B:])
    check_close(model.encoder_q[4].weight.grad, ref_sd["encoder_q.4.weight"].grad, 5e-3, "head weight gradient")
