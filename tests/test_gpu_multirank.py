"""world_size 2 ON HARDWARE with a single MI355X: both ranks run on cuda:0 (real HIP kernels,
real streams, the captured hipGraphs, DistributedDataParallel), collectives go over gloo (device
all-gather / all-to-all payloads staged through the host underneath torch.distributed's API, so the
product's own exchange code -- routed all-to-all (the default), all-gather, peer row pull -- is what runs).  RCCL itself
refuses two ranks on one device, so this is not the fabric path -- it is the W > 1 HOST SEQUENCING
executed with the product kernels instead of the ATen double, plus the peer row-pull exchange
(`COCLR_SHUFFLE=pull`, csrc/nce.hip pull_rows_kernel) through real hipIpc mappings between two
processes.  Checked against the two-rank fixtures recorded from the reference under DDP/gloo."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, kind, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # two processes on ONE GPU: equal stream priorities (see tests/bench_rehearse_gpu.py)
        os.environ.setdefault("COCLR_WGRAD_PRIORITY", "0")
        import sys
        here = os.path.dirname(os.path.abspath(__file__))
        for p in (here, os.path.dirname(here)):
            if p not in sys.path:
                sys.path.insert(0, p)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import coclr_amd.model.pretrain as impl
        import model.pretrain as product
        from _cases import build_model, case_inputs, check_close, load_golden, loss_fn

        # gloo carries device tensors for broadcast / all_reduce only: all_gather_into_tensor and
        # all_to_all_single of DEVICE tensors are staged through the host at the torch.distributed
        # level, so the product's own concat_all_gather / _routed_shuffle / _encode_keys code runs
        # unmodified (over RCCL these two calls are the native collectives)
        gather_native, a2a_native = dist.all_gather_into_tensor, dist.all_to_all_single

        def all_gather_into_tensor(out, tensor, *a, **kw):
            if not tensor.is_cuda:
                return gather_native(out, tensor, *a, **kw)
            host = torch.empty(out.shape, dtype=out.dtype)
            gather_native(host, tensor.contiguous().cpu(), *a, **kw)
            out.copy_(host)

        def all_to_all_single(out, tensor, output_split_sizes=None, input_split_sizes=None, *a, **kw):
            if not tensor.is_cuda:
                return a2a_native(out, tensor, output_split_sizes, input_split_sizes, *a, **kw)
            host = torch.empty(out.shape, dtype=out.dtype)
            a2a_native(host, tensor.contiguous().cpu(), output_split_sizes, input_split_sizes, *a, **kw)
            out.copy_(host)
        dist.all_gather_into_tensor = all_gather_into_tensor
        dist.all_to_all_single = all_to_all_single

        name = "%s_s3d_small_world2" % kind
        gold = load_golden("%s_rank%d" % (name, rank))
        cfg = gold["cfg"]
        B = cfg["B"]

        def run(mode):
            impl._SHUFFLE_MODE = mode
            model = build_model(cfg, product).cuda()
            ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0])
            opt = torch.optim.Adam([{"params": p} for _, p in ddp.named_parameters()], lr=1e-3,
                                   weight_decay=1e-5)
            ddp.train()
            if kind == "coclr":
                model.sampler.eval()
            outs = []
            for step, rec in enumerate(gold["steps"]):
                blocks, extra = case_inputs(cfg, step, world)
                sl = slice(rank * B, (rank + 1) * B)
                torch.manual_seed(cfg["perm_seed"] + step)
                if kind == "infonce":
                    out, tgt = ddp(blocks[0][sl].cuda())
                else:
                    out, tgt = ddp(blocks[0][sl].cuda(), blocks[1][sl].cuda(), extra[sl].cuda())
                loss = loss_fn(kind, out, tgt)
                opt.zero_grad()
                loss.backward()
                opt.step()
                outs.append(out.detach().clone())
                if step == 0:
                    check_close(out, rec["logits"], 1e-3, "rank %d logits (%s)" % (rank, mode))
                    if kind == "infonce":
                        assert torch.equal(tgt.cpu(), rec["target"])
                    else:
                        assert torch.equal(tgt.cpu().nonzero(), rec["target"])
                    sd = model.state_dict()
                    assert int(sd["queue_ptr"]) == int(rec["queue_ptr"])
                    bw = B * world
                    ptr0 = (int(sd["queue_ptr"]) - bw) % cfg["K"]
                    # keys of a key encoder that normalises over TWO clips per rank (8 values per
                    # channel in the last stage): the fixture itself moves by ~1e-3 under fp32
                    # re-association (tests/test_host_cpu.py); logits above are held to 1e-3
                    check_close(sd["queue"][:, ptr0:ptr0 + bw], rec["queue_cols"], 3e-3, "queue cols")
            assert opt._plan is not None           # the single-launch Adam ran under DDP
            return outs, {k: v.detach().clone() for k, v in model.state_dict().items()}

        outs_pull, sd_pull = run("pull")
        outs_ag, sd_ag = run("allgather")
        # the DEFAULT exchange (COCLR_SHUFFLE=routed: all_to_all_single of exactly the clips each
        # rank encodes, pretrain.py:_routed_shuffle) with the real kernels: first step only
        gold["steps"] = gold["steps"][:1]
        outs_rt, _ = run("routed")
        assert torch.equal(outs_rt[0], outs_ag[0]), "routed all-to-all and all-gather exchange disagree"
        # ... and the default, COCLR_SHUFFLE=auto: the first exchange runs routed AND pull, finds them
        # bit-identical on both ranks and leaves the process on the HIP row pull
        outs_auto, _ = run("auto")
        assert impl._SHUFFLE_MODE == "pull" and impl._SHUFFLE_INFO["selected"] == "pull", impl._SHUFFLE_INFO
        assert torch.equal(outs_auto[0], outs_ag[0]), "auto-selected exchange and all-gather exchange disagree"
        # the two exchange schemes deliver the same clips: the first step is bit-identical; later steps
        # are held to 5e-3 (the two runs see different allocator / graph-capture histories; nothing in
        # the kernels is order-dependent any more, but Adam at initialisation would turn a single
        # re-associated sum into +-lr weight changes under a 2-clip BatchNorm)
        assert torch.equal(outs_pull[0], outs_ag[0]), "row pull and all-gather exchange disagree"
        for a, b in zip(outs_pull[1:], outs_ag[1:]):
            check_close(a, b, 5e-3, "later steps, pull vs all-gather")
        for k in sd_pull:
            # (parameters after Adam steps are not comparable at this level: at initialisation the
            # update is ~lr*sign(g) and round-off flips signs, tests/test_host_cpu.py explains)
            if k.startswith("queue") or k.endswith(("running_mean", "running_var")):
                if sd_pull[k].is_floating_point():
                    check_close(sd_pull[k], sd_ag[k], 1e-2, k)      # after two noisy Adam steps
                else:
                    assert torch.equal(sd_pull[k], sd_ag[k]), k
        # replicas bit-identical across the two ranks
        for k in [k for k in sd_pull if k.startswith("queue")]:
            t = sd_pull[k].double().reshape(-1).cpu()
            dig = torch.stack([t.sum(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64)).sum()])
            got = [torch.zeros_like(dig) for _ in range(world)]
            dist.all_gather(got, dig)
            assert all(torch.equal(got[0], d) for d in got), k
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: %s\n%s" % (e, traceback.format_exc())))


@pytest.mark.parametrize("kind,port", [("infonce", 29731), ("coclr", 29732)])
def test_two_ranks_on_one_gpu_match_reference(kind, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, kind, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(60)
    for rank, msg in results:
        assert msg == "ok", "rank %d: %s" % (rank, msg)
