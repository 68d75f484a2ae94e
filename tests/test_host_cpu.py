"""CPU tier for the HOST side of the product (engine tape, inception wiring, step
sequencing of InfoNCE / UberNCE / CoCLR, DDP + gloo collectives at world_size 2) with the
kernel entry points replaced by the ATen test double in tests/fake_backend.py.  Numbers
are compared with the reference fixtures in tests/golden/.  The product itself refuses to
run on CPU tensors (test_product_has_no_cpu_path)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fake_backend
from _cases import (assert_checksums, build_model, case_inputs, check_close, checksum_table,
                    compare_state, compare_step, load_golden, loss_fn, sample)


@pytest.fixture
def fake(monkeypatch):
    fake_backend.install(monkeypatch)


def _run_steps(name, rank=0, world=1, wrap_ddp=False):
    import model.pretrain as product
    gold = load_golden(name if world == 1 else "%s_rank%d" % (name, rank))
    cfg = gold["cfg"]
    kind, B = cfg["kind"], cfg["B"]
    model = build_model(cfg, product)
    keys = gold["init_checksums"]["keys"]
    assert_checksums(checksum_table(model.state_dict(), keys), gold["init_checksums"]["vals"])
    wrapped = torch.nn.parallel.DistributedDataParallel(model) if wrap_ddp else model
    params = [{"params": p} for _, p in wrapped.named_parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)
    model.train()
    if kind == "coclr":
        model.sampler.eval()
    from oracle import coclr_oracle as orc
    for step, rec in enumerate(gold["steps"]):
        blocks, extra = case_inputs(cfg, step, world)
        sl = slice(rank * B, (rank + 1) * B)
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        torch.manual_seed(cfg["perm_seed"] + step)
        if kind == "infonce":
            out, tgt = wrapped(blocks[0][sl])
        elif kind == "ubernce":
            out, tgt = wrapped(blocks[0][sl], extra[sl])
        else:
            out, tgt = wrapped(blocks[0][sl], blocks[1][sl], extra[sl])
        loss = loss_fn(kind, out, tgt)
        opt.zero_grad()
        loss.backward()
        grads = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
        after = model.state_dict()
        assert int(after["queue_ptr"]) == int(rec["queue_ptr"])
        if step == 0:
            # The double re-associates some sums (BN statistics from partial sums, fused
            # affine), so it is ~1e-5..1e-4 off the reference in the forward; gradients at
            # initialisation are ill-conditioned (tests/_cases.compare_step).  Host-logic
            # mistakes show up as O(1) errors, which is what these bounds catch.
            check_close(out, rec["logits"], 5e-4, "logits")
            if kind == "infonce":
                assert torch.equal(tgt, rec["target"])
            else:
                assert torch.equal(tgt.nonzero(), rec["target"])
            check_close(loss, rec["loss"], 5e-3, "loss")
            for k, ref in rec["grads"].items():
                if os.environ.get("COCLR_TEST_VERBOSE"):
                    from _cases import rel_err
                    print("rank", rank, "grad", k, "rel err %.3e" % rel_err(sample(grads[k]), ref))
                # two-rank fixtures run BatchNorm over 2 clips per rank (8 values per channel
                # in the last stage): re-associating one convolution (e.g. the fused inception
                # heads vs three separate launches, identical per block to 7e-8) moves these
                # gradients by up to 0.4 of the tensor maximum, in the double and on the
                # reference alike; the GPU tier bounds gradients against a float64 evaluation
                check_close(sample(grads[k]), ref, 2e-1 if world == 1 else 6e-1, "grad " + k)
            compare_state(rec, after, B * world, cfg["K"], tol=1e-3)
        elif world == 1:
            # later steps: against the oracle continued from the product's own state (the
            # first Adam step turns round-off into +-2e-3 weight differences, so the
            # recorded trajectory cannot be followed tightly by ANY other build)
            sd = orc.training_state(before)
            pb = [blocks[0]] if kind != "coclr" else [(blocks[0], blocks[1])]
            (o_out, o_tgt), = orc.nce_step(sd, kind, cfg["network"], pb, [extra], cfg["dim"],
                                           cfg["K"], cfg["m"], cfg["T"], rec["perm"],
                                           topk=cfg.get("topk", 5),
                                           reverse=cfg.get("reverse", False))
            check_close(out, o_out, 5e-4, "step %d logits vs oracle" % step)
            assert torch.equal(tgt, o_tgt)
            for k, v in sd.items():
                if ".block" in k:
                    continue        # alias keys of the S3D stages: the oracle updates the named ones
                if k.startswith("queue") or k.endswith(("running_mean", "running_var")):
                    if v.is_floating_point():
                        check_close(after[k], v, 1e-3, "step %d %s" % (step, k))
                    else:
                        assert torch.equal(after[k], v), k
        opt.step()
    return model


@pytest.mark.parametrize("name", ["infonce_s3d_small", "ubernce_s3d_small", "coclr_s3d_small",
                                  "coclr_s3d_small_reverse_cold", "infonce_s3dg_small",
                                  "infonce_s3d_conditioned"])
def test_host_logic_matches_reference(fake, name):
    _run_steps(name)


def test_host_logic_r50(fake):
    _run_steps("infonce_r50_small")


def _ddp_worker(rank, world, name, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // world))
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class MP:   # minimal monkeypatch stand-in inside the spawned process
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        fake_backend.install(MP)
        model = _run_steps(name, rank=rank, world=world, wrap_ddp=True)
        # replicas must stay bit-identical: queues are built from the same gathered keys
        qsum = model.queue.double().sum().reshape(1)
        allq = [torch.zeros_like(qsum) for _ in range(world)]
        dist.all_gather(allq, qsum)
        assert all(torch.equal(allq[0], t) for t in allq)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("name,port", [("infonce_s3d_small_world2", 29701),
                                       ("coclr_s3d_small_world2", 29702)])
def test_two_rank_gloo_ddp_matches_reference(name, port):
    """world_size=2 over gloo: shuffle-BN all-gather + permutation broadcast, fused key
    gather, DDP gradient all-reduce -- against fixtures recorded from the reference under
    DDP/gloo with the same seeds."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, name, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_product_has_no_cpu_path():
    """Without the double, host tensors are rejected loudly (no oracle / ATen fallback)."""
    import model.pretrain as product
    from coclr_amd._lib import HipLibraryError
    torch.manual_seed(0)
    m = product.InfoNCE('s3d', 128, 32, 0.999, 0.07)
    with pytest.raises(HipLibraryError):
        m(torch.randn(2, 2, 3, 16, 64, 64))


def test_state_dict_is_reference_compatible():
    """Key set incl. the S3D alias keys, buffer dtypes and requires_grad flags
    (SURVEY.md 8b): 924 backbone keys per S3D encoder, 1858 for InfoNCE."""
    import model.pretrain as product
    torch.manual_seed(0)
    m = product.CoCLR('s3d', 128, 64, 0.999, 0.07, topk=5)
    sd = m.state_dict()
    for k in ("queue", "queue_ptr", "queue_second", "queue_vname", "queue_label",
              "encoder_q.0.Conv_1a.conv1.weight", "encoder_q.0.block1.0.conv1.weight",
              "encoder_k.0.Mixed_5c.branch3.1.bn.num_batches_tracked", "sampler.4.bias",
              "encoder_q.2.weight", "encoder_q.4.bias"):
        assert k in sd, k
    assert sd["queue"].shape == (128, 64) and sd["queue_ptr"].dtype == torch.long
    assert sd["encoder_q.0.Conv_1a.conv1.weight"].data_ptr() == \
        sd["encoder_q.0.block1.0.conv1.weight"].data_ptr()
    assert len([k for k in sd if k.startswith("encoder_q.0.")]) == 924
    flags = {n.split(".")[0]: p.requires_grad for n, p in m.named_parameters()}
    assert flags == {"encoder_q": True, "encoder_k": False, "sampler": False}
    assert sum(p.numel() for p in m.encoder_q.parameters()) == 9090848   # 9.09 M
    assert abs(float(sd["queue"].norm(dim=0).mean()) - 1.0) < 1e-5
    assert m.queue_is_full is False and m.topk == 5 and m.reverse is False
    # positional signature used by main_coclr.py:160
    product.CoCLR('s3d', 128, 64, 0.999, 0.07, 5, True)
    with pytest.raises(NotImplementedError):
        product.InfoNCE('resnet18')


@pytest.mark.parametrize("world,B", [(2, 3), (4, 8), (8, 32)])
def test_routed_shuffle_plan_delivers_every_clip_once(world, B):
    """The shuffle-BN exchange as an all-to-all (model.pretrain.route_plan): simulate every rank's
    send buffer and the all_to_all_single semantics, and check that each rank ends up with exactly
    the clips perm[r*B:(r+1)*B] in that order -- what the reference obtains by all-gathering
    everything and indexing (model/pretrain.py:106-124)."""
    from coclr_amd.model.pretrain import route_plan
    g = torch.Generator().manual_seed(world * 100 + B)
    for trial in range(5):
        perm = torch.randperm(world * B, generator=g).tolist()
        plans = [route_plan(perm, B, world, r) for r in range(world)]
        # sender side: rank s lays its clips out by destination
        sendbufs = [[s * B + li for li in plans[s][0]] for s in range(world)]     # global clip ids
        for s in range(world):
            assert sorted(plans[s][0]) == list(range(B))            # every local clip leaves once
            assert sum(plans[s][1]) == B
        for r in range(world):
            send_order, in_splits, out_splits, pos = plans[r]
            # all_to_all_single: chunk (src -> r) is the r-th chunk of src's send buffer
            recv = []
            for src in range(world):
                off = sum(plans[src][1][:r])
                chunk = sendbufs[src][off:off + plans[src][1][r]]
                assert len(chunk) == out_splits[src]                 # split sizes agree pairwise
                recv += chunk
            assert len(recv) == B
            assert [recv[p] for p in pos] == perm[r * B:(r + 1) * B]


def test_pack_plan_replays_recorded_relayouts(fake, monkeypatch):
    """engine.PackPlan: the first pass of a module launches its weight re-layouts one by one and
    records them; later passes replay them as one batch and skip the single launches (forward and
    backward operands alike); results do not change; moving a parameter re-records."""
    from coclr_amd import engine, ops
    from backbone.select_backbone import select_backbone
    torch.manual_seed(3)
    net, _ = select_backbone("s3d")
    mod = net.Mixed_3b
    x = torch.randn(2, 192, 4, 8, 8, requires_grad=True)
    calls = {"single": 0, "batch": 0}
    single, batch = ops.conv_pack_weights, ops.conv_pack_batch

    def count_single(*a, **k):
        calls["single"] += 1
        return single(*a, **k)

    def count_batch(*a, **k):
        calls["batch"] += 1
        return batch(*a, **k)

    monkeypatch.setattr(ops, "conv_pack_weights", count_single)
    monkeypatch.setattr(ops, "conv_pack_batch", count_batch)

    def step():
        for p in mod.parameters():
            p.grad = None
        x.grad = None
        y = engine.run_module(mod, x)
        y.square().mean().backward()
        return y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in mod.parameters()]

    y1, dx1, g1 = step()
    n_first = calls["single"]
    assert n_first > 0 and calls["batch"] == 0
    calls["single"] = 0
    y2, dx2, g2 = step()
    # the fake's batch replays the recorded requests through its own single-request routine,
    # which is not the patched name: no single launches are issued by the engine itself
    assert calls["batch"] == 1 and calls["single"] == 0
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    plan, = [p for p in mod.__dict__["_coclr_packplans"].values()]
    assert len(plan.requests) == n_first
    # a parameter that moved (e.g. module.to(...)) invalidates the table: that pass re-records
    w = next(mod.parameters())
    w.data = w.data.clone()
    calls.update(single=0, batch=0)
    y3, dx3, _ = step()
    assert calls["batch"] == 0 and calls["single"] == n_first
    assert torch.equal(y1, y3) and torch.equal(dx1, dx3)
    calls.update(single=0, batch=0)
    step()
    assert calls["batch"] == 1 and calls["single"] == 0


def test_winograd_policy():
    """ops.conv_geom(): which convolutions go through the Winograd kernels (capability vs policy)."""
    from coclr_amd import ops
    if not (ops.WINOGRAD and ops.WINOGRAD_HW):
        pytest.skip("Winograd switched off by the environment")
    # temporal halves of the separable units: always, from 16 input channels up
    assert ops.conv_geom(4, 64, 64, (8, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)).algo == 1
    assert ops.conv_geom(4, 8, 64, (8, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)).algo == 0
    assert ops.conv_geom(4, 64, 64, (8, 8, 8), (3, 1, 1), (2, 1, 1), (1, 0, 0)).algo == 0
    # spatial halves: 16x16 maps and up, even extents, stride 1
    hw = 1
    assert ops.conv_geom(4, 64, 192, (4, 32, 32), (1, 3, 3), (1, 1, 1), (0, 1, 1)).algo == hw
    assert ops.conv_geom(4, 96, 128, (4, 16, 16), (1, 3, 3), (1, 1, 1), (0, 1, 1)).algo == hw
    assert ops.conv_geom(4, 96, 208, (4, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1)).algo == 0
    assert ops.conv_geom(4, 96, 208, (4, 17, 17), (1, 3, 3), (1, 1, 1), (0, 1, 1)).algo == 0
    assert ops.conv_geom(4, 64, 64, (4, 32, 32), (1, 3, 3), (1, 2, 2), (0, 1, 1)).algo == 0
    # the kernels themselves accept the small even maps (capability), and the data gradient of a
    # Winograd conv is a Winograd conv
    g = ops.ConvGeom(2, 32, 48, (3, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), algo=1)
    assert g.dgrad().algo == 1 and g.dgrad().Cin == 48 and g.dgrad().Cout == 32
    with pytest.raises(ValueError):
        ops.ConvGeom(2, 32, 48, (3, 7, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), algo=1)
    with pytest.raises(ValueError):
        ops.ConvGeom(2, 32, 48, (3, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0), algo=3)
    with pytest.raises(ValueError):
        ops.ConvGeom(2, 32, 48, (3, 8, 8), (1, 3, 3), (1, 1, 1), (0, 1, 1), algo=2)
    # round 6: F(4,3) (algo 2) for the temporal layers from 16 frames up; the stride-2 temporal stem conv in
    # polyphase form (algo 1 on that stencil) at the benchmark clip length, its data gradient in two Winograd
    # phases that write through a lattice along T
    if ops.WINOGRAD_T4:
        assert ops.conv_geom(4, 192, 192, (16, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)).algo == 2
        assert ops.conv_geom(4, 192, 192, (16, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0)).dgrad().algo == 2
        assert ops.ConvGeom(2, 32, 48, (3, 8, 8), (3, 1, 1), (1, 1, 1), (1, 0, 0), algo=2).algo == 2    # capability
    if ops.WINOGRAD_POLY7:
        stem = ops.conv_geom(4, 64, 64, (32, 16, 16), (7, 1, 1), (2, 1, 1), (3, 0, 0))
        assert stem.algo == 1 and stem.dgrad().algo == 0
        assert ops.conv_geom(4, 64, 64, (16, 16, 16), (7, 1, 1), (2, 1, 1), (3, 0, 0)).algo == 0
        with pytest.raises(ValueError):
            ops.ConvGeom(4, 64, 64, (31, 16, 16), (7, 1, 1), (2, 1, 1), (3, 0, 0), algo=1)
        if ops.WINOGRAD_T4 and ops.WINOGRAD_PHASES:
            phases = stem.dgrad_phases()
            assert sorted((nk, pg.algo) for pg, _, nk, _ in phases) == [(3, 2), (4, 2)]
            assert all(pg.lattice[0][0] == 2 for pg, _, _, _ in phases)


def test_forward_without_backward_releases_the_tape(fake):
    """main_coclr.py:403 skips loss.backward() until the queue is full: a grad-enabled forward
    that is never differentiated must not keep its activation tape alive (the tape used to hold
    the very tensor autograd stamps with grad_fn -- an uncollectable cycle)."""
    import gc
    import weakref
    from coclr_amd import engine
    from backbone.select_backbone import select_backbone
    torch.manual_seed(0)
    net, _ = select_backbone('s3d')
    net.train()
    live = []
    orig_init = engine.Run.__init__

    def tracking_init(self, *a, **kw):
        orig_init(self, *a, **kw)
        live.append(weakref.ref(self))

    engine.Run.__init__ = tracking_init
    gc.collect()
    gc.disable()         # plain reference counting must be enough: no cycles through the tape
    try:
        for _ in range(3):
            y = net(torch.randn(2, 3, 8, 32, 32))
            assert y.requires_grad and y.grad_fn is not None
            del y
        alive = [r for r in live if r() is not None]
        assert len(live) >= 3 and not alive, "%d of %d runs still alive" % (len(alive), len(live))
        # and a differentiated pass still works through the fresh output alias
        y = net(torch.randn(2, 3, 8, 32, 32))
        y.sum().backward()
        assert net.Conv_1a.conv1.weight.grad is not None
    finally:
        gc.enable()
        engine.Run.__init__ = orig_init


def test_resumed_queue_pointer_is_validated(fake):
    """A checkpoint written with another global batch leaves queue_ptr off the new batch grid:
    the reference raises on the slice assignment (model/pretrain.py:93); the device-side enqueue
    must not write past the queue (it used to clobber the neighbouring buffers of the flat
    allocation)."""
    import model.pretrain as product
    torch.manual_seed(0)
    m = product.InfoNCE('s3d', 128, 32, 0.999, 0.07)
    m.train()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["queue_ptr"][0] = 30           # K=32, batch 4: 30 + 4 > 32
    m.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="queue_ptr"):
        m(torch.randn(4, 2, 3, 8, 32, 32))
    sd["queue_ptr"][0] = 28
    m.load_state_dict(sd)
    m(torch.randn(4, 2, 3, 8, 32, 32))
    assert int(m.queue_ptr) == 0


def _multi_rank_worker(rank, world, kind, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class MP:
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        fake_backend.install(MP)
        import coclr_amd.model.pretrain as impl
        import model.pretrain as product
        B, K, clip = 2, 64, (3, 8, 32, 32)
        assert K % (B * world) == 0

        def run(routed):
            impl._SHUFFLE_MODE = "routed" if routed else "allgather"
            torch.manual_seed(0)
            if kind == "infonce":
                model = product.InfoNCE('s3d', 128, K, 0.999, 0.07)
            else:
                model = product.CoCLR('s3d', 128, K, 0.999, 0.07, topk=5)
                g = torch.Generator().manual_seed(7)
                model.queue_label.fill_(1)
                model.queue_vname.copy_(torch.randint(0, 6, (K,), generator=g))
            ddp = torch.nn.parallel.DistributedDataParallel(model)
            opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3,
                                   weight_decay=1e-5)
            ddp.train()
            if kind == "coclr":
                model.sampler.eval()
            outs = []
            for step in range(2):
                g = torch.Generator().manual_seed(50 + step)
                blocks = [torch.randn(B * world, 2, *clip, generator=g) for _ in range(2)]
                vsrc = torch.randint(0, 6, (B * world,), generator=g)
                sl = slice(rank * B, (rank + 1) * B)
                torch.manual_seed(900 + step)              # same permutation on every rank / scheme
                if kind == "infonce":
                    out, tgt = ddp(blocks[0][sl])
                    loss = torch.nn.functional.cross_entropy(out, tgt)
                else:
                    out, mask = ddp(blocks[0][sl], blocks[1][sl], vsrc[sl])
                    loss = (- torch.log((torch.softmax(out, dim=1) * mask).sum(1))).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
                outs.append(out.detach().clone())
            sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
            return outs, sd

        outs_r, sd_r = run(True)
        outs_a, sd_a = run(False)
        # the routed all-to-all delivers exactly the clips the reference's all-gather + index keeps
        for a, b in zip(outs_r, outs_a):
            assert torch.equal(a, b), "routed vs all-gather logits differ"
        for k in sd_r:
            assert torch.equal(sd_r[k], sd_a[k]), "routed vs all-gather state differs: " + k
        # replicas stay bit-identical: queues (all of them), pointer, BN buffers after the broadcast
        for k in [k for k in sd_r if k.startswith("queue")] + ["encoder_q.4.bias"]:
            t = sd_r[k].double().reshape(-1)
            digest = torch.stack([t.sum(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum()])
            got = [torch.zeros_like(digest) for _ in range(world)]
            dist.all_gather(got, digest)
            assert all(torch.equal(got[0], d) for d in got), "replicas diverged in " + k
        assert int(sd_r["queue_ptr"]) == (2 * B * world) % K
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("world,kind,port", [(4, "infonce", 29711), (8, "infonce", 29712),
                                             (4, "coclr", 29713), (8, "coclr", 29714)])
def test_four_and_eight_rank_gloo(world, kind, port):
    """world_size 4 and 8 over gloo (kernels replaced by the ATen double): two DDP training steps of
    InfoNCE / CoCLR with the routed shuffle-BN exchange and with the reference's all-gather scheme
    -- bit-identical logits and state between the two, queues / pointer / parameters bit-identical
    across ranks.  The only multi-rank evidence available without an 8-GPU node."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_multi_rank_worker, args=(r, world, kind, port, q))
             for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _queue_sync_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class MP:
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        fake_backend.install(MP)
        import coclr_amd.model.pretrain as impl
        import model.pretrain as product
        from coclr_amd import parallel
        B, K, clip = 2, 64, (3, 8, 32, 32)

        def run(sync_queues):
            impl._SYNC_QUEUES = sync_queues
            torch.manual_seed(0)
            model = product.CoCLR('s3d', 128, K, 0.999, 0.07, topk=5)
            # every rank starts from ITS OWN random queues (an unseeded launch script) ...
            g = torch.Generator().manual_seed(100 + rank)
            model.queue.copy_(torch.nn.functional.normalize(torch.randn(128, K, generator=g), dim=0))
            model.queue_second.copy_(torch.nn.functional.normalize(torch.randn(128, K, generator=g), dim=0))
            model.queue_label.fill_(1)
            model.queue_vname.copy_(torch.randint(0, 6, (K,), generator=torch.Generator().manual_seed(7)))
            ddp = torch.nn.parallel.DistributedDataParallel(model)
            opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3, weight_decay=1e-5)
            ddp.train()
            model.sampler.eval()
            sizes, outs = [], []
            for step in range(3):
                g = torch.Generator().manual_seed(50 + step)
                blocks = [torch.randn(B * world, 2, *clip, generator=g) for _ in range(2)]
                vsrc = torch.randint(0, 6, (B * world,), generator=g)
                sl = slice(rank * B, (rank + 1) * B)
                torch.manual_seed(900 + step)
                parallel.TIMINGS = []
                out, mask = ddp(blocks[0][sl], blocks[1][sl], vsrc[sl])
                rows, parallel.TIMINGS = parallel.TIMINGS, None
                sizes.append(sum(nb for name, nb, _ in rows if "flat float32 buffer" in name))
                loss = (- torch.log((torch.softmax(out, dim=1) * mask).sum(1))).mean()
                opt.zero_grad()
                loss.backward()
                opt.step()
                outs.append(out.detach().clone())
                if step == 1 and rank == 0:
                    # ... and a state dict loaded mid-run ON RANK 0 ALONE (legal under the reference: DDP re-sends
                    # rank 0's buffers with every forward) makes the next forward send everything again ON EVERY
                    # RANK: the ranks agree on the message size over the host channel first (a rank sizing the
                    # broadcast from its own flag would post a different count than its peers)
                    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
                    sd0["queue"] = sd0["queue"].flip(1)
                    model.load_state_dict(sd0)
            return sizes, outs, {k: v.detach().clone() for k, v in model.state_dict().items()}

        sizes, outs, sd = run(False)
        sizes_ref, outs_ref, sd_ref = run(True)
        qbytes = 2 * 128 * K * 4
        assert sizes_ref[0] == sizes_ref[1] == sizes_ref[2] and sizes_ref[0] > qbytes, sizes_ref
        # first forward: everything (the ranks' queues differ); second: the BatchNorm statistics only; third:
        # everything again (load_state_dict in between)
        assert sizes[0] == sizes_ref[0] and sizes[2] == sizes_ref[0], (sizes, sizes_ref)
        assert sizes[1] == sizes_ref[0] - qbytes, (sizes, sizes_ref)
        # same results as re-sending the queues with every forward (the reference's DDP broadcast_buffers)
        for a, b in zip(outs, outs_ref):
            assert torch.equal(a, b)
        for k in sd:
            assert torch.equal(sd[k], sd_ref[k]), k
        # rank 0's queues won on every rank (BatchNorm statistics are rank-local until the next forward's broadcast)
        for k in ("queue", "queue_second", "queue_ptr", "queue_vname"):
            t = sd[k].double().reshape(-1)
            digest = torch.stack([t.sum(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum()])
            got = [torch.zeros_like(digest) for _ in range(world)]
            dist.all_gather(got, digest)
            assert all(torch.equal(got[0], d) for d in got), "replicas diverged in " + k
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_queues_travel_with_the_first_broadcast_only():
    """DDP(broadcast_buffers=True) re-sends the queues from rank 0 with every forward (main_nce.py:172).  Here
    they travel with the first broadcast after construction / load_state_dict -- when ranks may hold different
    random queues -- and not again (every rank enqueues the same gathered keys at the same pointer): the
    steady-state broadcast is the BatchNorm statistics.  Results identical to re-sending them every time."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_queue_sync_worker, args=(r, 2, 29745, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def _bench_dry_run(world, port, extra_env=None, extra_args=(), bare=False):
    """bench.py exactly as the driver launches it (torch.distributed.run, one process per rank) on the host
    with the ATen double; returns (the ONE JSON line rank 0 printed, stderr).
    bare: no launcher -- a plain `python <script> --gpus N`, which has to start the N ranks itself
    (bench.self_launch)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    launcher = [] if bare else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                                "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd = [sys.executable] + launcher + [
        os.path.join(root, "tests", "bench_dryrun.py"), "--gpus", str(world), "--steps", "2",
        "--warmup", "1"] + list(extra_args)
    env = dict(os.environ, OMP_NUM_THREADS="1", COCLR_QUIET="1", **(extra_env or {}))
    if bare:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out.stderr


@pytest.mark.parametrize("world", [1, 2, 8])
def test_bench_launch_contract_dry_run(world):
    """One JSON line from rank 0 with the contract's fields, whole-job value = B*world*steps/time, K = 16384
    and weak scaling at world > 1.
    World 8: a fault is injected into the bucket hook (a bucket all-reduced with a wrong gradient in it --
    identical on every rank, so replicas still agree): the self-check must see it and the bench must end, with a
    valid value, on the rung that switches the hook off.
    World 2 (16 clips per rank, so that the OPTIONAL host-floor leg after the timed region runs): rank 1 never comes
    back from that leg -- the watchdog must print the record that is already complete, flagged
    `optional_leg_hung`, and every process must exit 0: a hang after the measurement cannot cost the measurement.
    World 2 is also started BARE -- `python <script> --gpus 2`, no torch.distributed.run around it, the way the
    round-5 driver typed its bench command: bench.py has to start the two ranks itself (bench.self_launch)."""
    fault = {"COCLR_BENCH_FAULT": "hook"} if world == 8 else \
        ({"COCLR_BENCH_FAULT": "floor_hangs"} if world == 2 else None)
    extra = ("--batch", "16", "--moco-k", "64", "--hang-timeout", "6") if world == 2 else ()
    rec, err = _bench_dry_run(world, 29720 + world, fault, extra, bare=world == 2)
    assert ("starting the ranks" in err) == (world == 2)
    B = 16 if world == 2 else 2
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in rec, key
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["scaling"] == "weak" and rec["higher_is_better"] is True and rec["dtype"] == "fp32"
    assert rec["config"]["global_batch"] == B * world and "DRY RUN" in rec["data"]
    assert ("moco-k=16384" in rec["config"]["workload"]) == (world == 8)
    assert ("moco-k=64 " in rec["config"]["workload"]) == (world == 2)
    assert abs(rec["value"] - B * world * 2 / (rec["ms_per_step"] * 2e-3)) <= 0.02 * rec["value"]
    # the fast-vs-serial bit-identity check of one step runs at every world size
    sc = rec["self_check"]
    assert sc["passed"] is True and sc["tensors_compared"] > 1000
    assert sc["trials"][0]["rung"] == 0
    if world == 1:
        assert "multi_gpu" not in rec and sc["rung"] == 0
        return
    # first-contact evidence: one checked step with cross-rank digests, then the
    # per-collective wall times of two instrumented steps
    mg = rec["multi_gpu"]
    assert mg["shuffle_mode"] == "routed" and mg["split_stages"] is True
    assert mg["shuffle_selection"]["requested"] == "auto"        # host tensors: nothing to pull
    assert mg["cross_rank"]["replicas_identical"] is True, mg["cross_rank"]
    assert mg["cross_rank"]["logits_finite_on_every_rank"] is True
    assert {"queue", "queue_ptr", "encoder_q.parameters", "encoder_k.parameters"} <= set(mg["cross_rank"]["fields"])
    names = " | ".join(c["collective"] for c in mg["collectives"])
    # (on rung >= 2 the bucket all-reduce is DDP's own and does not pass the named choke point)
    for what in ("all_to_all_single", "all_gather_into_tensor", "broadcast of the flat float32 buffer",
                 "host broadcast of the permutation") + (("ddp bucket 0 all_reduce",) if world != 8 else ()):
        assert what in names, (what, names)
    assert all(c["calls_per_step"] == 1.0 and c["ms_per_call"] > 0 for c in mg["collectives"])
    assert mg["attempts"] == [{"attempt": 0, "started_on_rung": 0, "rung_name": "fast", "ok": True,
                               "exit_codes": [0] * world}]
    if world == 8:
        # the injected race: rungs 0 and 1 differ from the serial step, rung 2 (no hook) matches, and THAT is timed
        assert mg["rung"] == 2 and mg["rung_name"] == "-hook" and sc["rung"] == 2
        assert [t["bit_identical_to_serial_on_every_rank"] for t in sc["trials"]] == [False, False, True]
        assert mg["cross_rank"]["replicas_identical"] is True      # ... which replicas-identical cannot see
    else:
        assert mg["rung"] == 0 and len(sc["trials"]) == 1
        hung = rec["optional_leg_hung"]
        assert "host floor" in hung["phase"] and "host_floor_ms_per_step" not in mg


def test_bench_dry_run_coclr_self_check():
    """BASELINE config 4's model through bench.py at N = 2: the self-check covers three encoders, four queues
    and the frozen sampler's EVAL mode (main_coclr.py:363) -- a bench that flips it (as `.train()` on the
    check's second DDP wrapper once did) makes every later step differ from the checked one."""
    rec, _ = _bench_dry_run(2, 29729, None, ("--model", "coclr"))
    sc, mg = rec["self_check"], rec["multi_gpu"]
    assert "CoCLR" in rec["metric"] and rec["value"] > 0
    assert sc["passed"] is True and sc["rung"] == 0 and mg["rung"] == 0, sc["trials"]
    assert sc["tensors_compared"] > 2000 and len(sc["trials"]) == 1


@pytest.mark.parametrize("fault,rung,attempts", [("routed_raises", 4, 2), ("routed_hangs", 4, 2)])
def test_bench_degradation_ladder(fault, rung, attempts):
    """a fault at each kind of rung, and bench.py still finishes with ONE valid line
    whose `multi_gpu.rung` names what had to be switched off.
      (wrong RESULTS tied to a switch are found by the self-check and walked down IN PROCESS: the bucket-hook
       fault rides on the W = 8 dry run above -- rung 2; a gradient lost while joins are deferred -- rung 1 --
       and keys off by 1e-3 under graph replay -- rung 5 -- on the one-GPU rehearsal with the real streams,
       tests/test_gpu_bench_rehearsal.py)
      routed_raises    the backend refuses all_to_all_single on every rank: the attempt ends, the supervisors
                       read WHICH exchange failed and start a new set of processes on the all-gather rung
      routed_hangs     one rank never enters the exchange: the watchdog names it after --hang-timeout and
                       ends the attempt; same recovery"""
    rec, err = _bench_dry_run(2, 29750 + rung + attempts, {"COCLR_BENCH_FAULT": fault},
                              ("--hang-timeout", "10"))      # 6 s flaked once on a loaded 8-core host
    mg = rec["multi_gpu"]
    assert rec["value"] > 0 and rec["n_gpus"] == 2
    assert mg["rung"] == rung, (mg["rung"], mg["attempts"], rec["self_check"]["trials"])
    assert rec["self_check"]["passed"] is True
    assert len(mg["attempts"]) == attempts and mg["attempts"][-1]["ok"] is True
    if attempts == 2:
        first = mg["attempts"][0]
        assert first["ok"] is False and first["started_on_rung"] == 0
        assert "all_to_all_single" in first["failed_ranks"][0]["last_collective"]
        assert mg["attempts"][1]["started_on_rung"] == 4 and mg["switches"]["shuffle"] == "allgather"
        assert "bench supervisor: attempt 0 on rung 0 (fast) failed" in err
    else:
        assert [t["bit_identical_to_serial_on_every_rank"] for t in rec["self_check"]["trials"]] == \
            [False] * rung + [True]


def test_bench_watchdog_names_the_stuck_collective():
    """A rank that stops making progress reports the exchange it issued last and ends the job with the
    contract's JSON line (value null) instead of hanging until the driver's timeout."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import time, bench\n"
            "from coclr_amd import parallel\n"
            "parallel.LAST[0], parallel.LAST[2] = 'all_to_all_single of 32 key clips (test)', 7\n"
            "d = bench.Watchdog(0, 8, 0.4, {'metric': 'clips/sec (whole node)', 'n_gpus': 8})\n"
            "d.at('timed steps')\n"
            "time.sleep(60)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode == 5, (out.returncode, out.stderr[-2000:])
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert rec["value"] is None and rec["n_gpus"] == 8
    assert rec["hang"]["last_collective"].startswith("all_to_all_single") and rec["hang"]["phase"] == "timed steps"
    assert "bench watchdog" in out.stderr


def _pull_fallback_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.set_num_threads(1)
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class MP:
            @staticmethod
            def setattr(obj, name, val):
                setattr(obj, name, val)
        fake_backend.install(MP)
        import warnings
        import torch.multiprocessing.reductions as red
        import coclr_amd.model.pretrain as impl
        import model.pretrain as product
        if rank == 1:
            def refuse(t):
                raise RuntimeError("hipIpcGetMemHandle: invalid argument (simulated)")
            red.reduce_tensor = refuse
        B, K, clip = 2, 64, (3, 8, 32, 32)

        def run(mode):
            impl._SHUFFLE_MODE = mode
            torch.manual_seed(0)
            model = product.InfoNCE('s3d', 128, K, 0.999, 0.07)
            model.train()
            g = torch.Generator().manual_seed(50)
            block = torch.randn(B * world, 2, *clip, generator=g)[rank * B:(rank + 1) * B]
            torch.manual_seed(900)
            with torch.no_grad():
                out, _ = model(block)
            return out

        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            got = run("pull")
        assert impl._SHUFFLE_MODE == "routed", "the fallback must switch every rank to the routed exchange"
        assert any("every rank uses the routed" in str(w.message) for w in caught)
        ref = run("routed")
        assert torch.equal(got, ref)
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


def test_pull_shuffle_falls_back_to_routed_on_every_rank():
    """COCLR_SHUFFLE=pull where ONE rank cannot export its staging buffers (hipIpc refused): all ranks
    agree over the host channel and fall back to the routed all-to-all together -- no rank is left
    waiting in a collective the others never enter."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_pull_fallback_worker, args=(r, 2, 29741, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)


def test_lockstep_emission_of_the_branch_tails_matches_sequential(fake, monkeypatch):
    """engine.drive_pair (the two separable branches of an inception block as coroutines in lockstep,
    their backward closures paired on the tape) against one unit after the other: same launches in a
    different order, identical results -- forward, every gradient, BatchNorm buffers (host logic; the
    fused kernels themselves are held to bit-identity on the GPU, tests/test_gpu_multi.py)."""
    from coclr_amd import engine
    from coclr_amd.backbone import s3dg
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 480, 4, 8, 8, generator=g).relu()
    dout = torch.randn(2, 512, 4, 8, 8, generator=g)
    results = []
    for pair in (True, False):
        monkeypatch.setattr(engine, "PAIR_UNITS", pair)
        torch.manual_seed(0)
        m = s3dg.SepInception(480, [192, 96, 208, 16, 48, 64]).train()
        xg = x.clone().requires_grad_(True)
        out = m(xg)
        out.backward(dout)
        results.append((out.detach().clone(), xg.grad.clone(),
                        {k: p.grad.clone() for k, p in m.named_parameters()},
                        {k: v.clone() for k, v in m.named_buffers()}))
    a, b = results
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(p is not None for p in a[2].values()) and len(a[2]) == 24
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_bn_backward_sums_formed_by_the_writing_data_gradient(fake, monkeypatch):
    """engine: when ONE data gradient writes a unit's d(activation) in one piece, it is asked to form that
    unit's BatchNorm backward sums (conv_fwd_multi bwd_bn) and the unit's own backward skips its reduction
    pass (bn_act_backward_multi partials) -- strided phases included; an activation with two gradient
    writers keeps the reduction.  Host logic on the CPU double, against the same model with the fusion
    switched off; the kernels are held to the reduction pass on the GPU (tests/test_gpu_multi.py)."""
    from coclr_amd import engine, ops
    from coclr_amd.backbone import s3dg
    monkeypatch.setattr(ops, "SMALL_CHANNEL", 0)        # every layer counts as "large"
    seen = {"bwd_bn": 0, "partials": 0, "two": 0}
    real_multi, real_bn = ops.conv_fwd_multi, ops.bn_act_backward_multi

    def spy_conv(calls):
        seen["bwd_bn"] += sum(1 for c in calls if c.get("bwd_bn") is not None)
        return real_multi(calls)

    def spy_bn(units):
        for u in units:
            if u.get("partials"):
                seen["partials"] += 1
                seen["two"] += len(u["partials"]) == 2
        return real_bn(units)

    monkeypatch.setattr(ops, "conv_fwd_multi", spy_conv)
    monkeypatch.setattr(ops, "bn_act_backward_multi", spy_bn)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 8, 32, 32, generator=g)
    results = []
    for fuse in (True, False):
        monkeypatch.setattr(engine, "FUSE_BN_REDUCE", fuse)
        torch.manual_seed(0)
        m = s3dg.S3D(input_channel=3).train()
        for k in seen:
            seen[k] = 0
        out = m(x)
        out.square().mean().backward()
        results.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                        dict(seen)))
    (oa, ga, sa), (ob, gb, sb) = results
    assert sb["bwd_bn"] == 0 and sb["partials"] == 0
    # Conv_1a.bn1 (two strided phases), Conv_2c.bn1 and the conv1 BatchNorms of the separable branches of the
    # nine blocks whose temporal data gradient has the epilogue
    assert sa["partials"] >= 2 + 2 * 9 - 4 and sa["two"] == 1, sa
    assert sa["bwd_bn"] == sa["partials"] + 1, sa
    assert torch.equal(oa, ob)
    for k in ga:
        ref = gb[k]
        assert (ga[k] - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-8, k


def test_stem_backward_takes_the_short_form(fake, monkeypatch):
    """engine: the unit whose data gradient nobody needs (Conv_1a.conv1 over the clip) skips BatchNorm's backward
    apply pass -- reduction + coefficients, then the weight gradient that applies them (ops.conv_wgrad_bn) -- and
    every other unit keeps the two-pass form; COCLR_WGRAD_BN=0 switches it off.  Host logic on the CPU double,
    against the same model without it; the kernel is held bit-identical to the two-pass form on the GPU
    (tests/test_gpu_kernels.py::test_stem_weight_gradient_applies_batchnorm_backward)."""
    from coclr_amd import engine, ops
    from coclr_amd.backbone import s3dg
    seen = {"short": 0, "coeffs": 0}
    real_w, real_c = ops.conv_wgrad_bn, ops.bn_act_backward_coeffs

    def spy_w(geom, *a, **k):
        seen["short"] += 1
        assert geom.k == (1, 7, 7) and geom.Cin == 3
        return real_w(geom, *a, **k)

    def spy_c(*a, **k):
        seen["coeffs"] += 1
        return real_c(*a, **k)

    monkeypatch.setattr(ops, "conv_wgrad_bn", spy_w)
    monkeypatch.setattr(ops, "bn_act_backward_coeffs", spy_c)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 3, 8, 32, 32, generator=g)
    results = []
    for on in (True, False):
        monkeypatch.setattr(engine, "WGRAD_BN", on)
        torch.manual_seed(0)
        m = s3dg.S3D(input_channel=3).train()
        seen["short"] = seen["coeffs"] = 0
        out = m(x)
        out.square().mean().backward()
        results.append((out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}, dict(seen)))
    (oa, ga, sa), (ob, gb, sb) = results
    assert sa == {"short": 1, "coeffs": 1} and sb == {"short": 0, "coeffs": 0}, (sa, sb)
    assert torch.equal(oa, ob)
    for k in ga:
        ref = gb[k]
        assert (ga[k] - ref).abs().max() <= 1e-5 * ref.abs().max() + 1e-8, k


def test_launch_plan_mechanics():
    """coclr_amd/plan.py without a GPU: a log of (C function, frozen arguments) and Python callables is re-issued in
    order; byref'd descriptors are COPIED at recording time (the shared geometry descriptors are rewritten by later
    calls); top-level pointers into a moved input are found and patched, pointers embedded in by-pointer tables are
    reported (the caller then does not follow a moved input); host-side queries are never logged."""
    import ctypes as C
    from coclr_amd import plan, _lib

    class Desc(C.Structure):
        _fields_ = [("n", C.c_int32), ("ptr", C.c_void_p)]

    seen = []
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(Desc), C.c_void_p, C.c_int32, C.c_void_p)

    def impl(d, p, k, stream):
        seen.append((d.contents.n, d.contents.ptr, p, k, stream))
        return 0
    fn = CB(impl)
    fn.argtypes = [C.POINTER(Desc), C.c_void_p, C.c_int32, C.c_void_p]

    class FakeLib:
        coclr_conv3d_fwd = fn
        coclr_conv3d_ntiles = fn           # a query name: passed through, never logged

    rec = plan.Recorder(stream=7)
    rec.proxy = plan._Proxy(rec, FakeLib())
    shared = Desc(3, 0x5000)
    rec.proxy.coclr_conv3d_fwd(C.byref(shared), 0x1000 + 64, 5, 7)
    shared.n = 99                                   # the shared descriptor is rewritten by the next call
    rec.py(lambda: seen.append("dependency"))
    rec.proxy.coclr_conv3d_fwd(C.byref(shared), 0x9000, 6, 7)
    rec.proxy.coclr_conv3d_ntiles(C.byref(shared), None, 0, None)
    p = rec.plan
    assert p.ncalls == 2 and len(p.entries) == 3 and p.stream == 7
    del seen[:]
    before = _lib.CALLS[0]
    p.replay()
    assert seen == [(3, 0x5000, 0x1040, 5, 7), "dependency", (99, 0x5000, 0x9000, 6, 7)]
    assert _lib.CALLS[0] == before + 2
    # the input that lived at [0x1000, 0x2000) moved to 0x7000: one top-level reference, offset 64
    refs = p.pointer_refs(0x1000, 0x2000)
    assert refs == [(0, 1, 64)]
    p.patch(refs, 0x7000)
    del seen[:]
    p.replay()
    assert seen[0] == (3, 0x5000, 0x7040, 5, 7)
    # an address inside a struct handed over by pointer is not patched -- it is reported
    assert p.embedded_refs(0x5000, 0x5008) and not p.embedded_refs(0x6000, 0x6100)
    # a failing call raises with the entry's name
    bad = CB(lambda d, q, k, s: 719)
    bad.argtypes = fn.argtypes
    p.entries.append((bad, [None, None, C.c_int32(0), None]))
    with pytest.raises(_lib.HipLibraryError):
        p.replay()
