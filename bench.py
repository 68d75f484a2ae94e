#!/usr/bin/env python
"""Throughput benchmark of the CoCLR training hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's train_one_epoch body (main_nce.py:307-331)
on one synthetic batch already resident in HBM:
    logits, labels = DDP(InfoNCE('s3d'))(block)      q fwd, momentum update, shuffle-BN,
                                                     k fwd, logits, enqueue (+ collectives)
    loss = CrossEntropyLoss(logits, labels); optimizer.zero_grad(); loss.backward();
    optimizer.step()                                 Adam lr 1e-3 wd 1e-5
Metric (BASELINE.json): clips/sec over the whole job = B * world / step time, B = 32
clips of 3x32x128x128 per GPU, fp32.  Nothing is skipped inside the timed region.
Caller-side work that is not on the hot path (accuracy meters and their .item() syncs,
tqdm, TensorBoard) is left out and the optimiser is torch's fused Adam over one param
group -- both are noted in `config`.

Prints ONE JSON line on rank 0 (see README/DESIGN for the field contract), including
  roofline     -- the dominant kernel (conv implicit GEMM of Conv_2c.conv1, fp32 MFMA),
                  timed live with HIP events on the launch stream inside the timed region
  cpu_baseline -- the CPU oracle (port of the reference step) on this host's cores, N=1 only
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md, v_mfma_f32_32x32x2_f32
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU")
    ap.add_argument("--moco-k", type=int, default=None)
    ap.add_argument("--net", default="s3d")
    ap.add_argument("--model", default="infonce", choices=["infonce", "coclr"])
    ap.add_argument("--seq-len", type=int, default=32)
    ap.add_argument("--img-dim", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="intra-op threads of the CPU baseline (16 is the fastest on the 128-core "
                         "GPU host: 8->2.67, 16->2.88, 32->2.77, 64->1.67, 128->0.78 clips/s)")
    return ap.parse_args()


def synthetic_block(B, seq_len, img_dim, device, seed):
    """ToTensor-like U[0,1) frames -> Normalize(channel=1) -> (B,2,3,T,H,W), the recipe of
    main_nce.py:207-209,299-302, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    x = torch.rand(B, 3, 2 * seq_len, img_dim, img_dim, generator=g, device=device)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1, 1)
    x = (x - mean) / std
    return x.view(B, 3, 2, seq_len, img_dim, img_dim).transpose(1, 2).contiguous()


class KernelTimer:
    """HIP-event timing of selected conv launches on the stream they are enqueued on
    (our kernels run on torch's current stream, so torch.cuda.Event brackets them)."""

    def __init__(self, match):
        self.match = match
        self.events = []
        self.enabled = False

    def install(self):
        from coclr_amd import ops
        inner = ops.conv_fwd
        timer = self

        def timed_conv_fwd(geom, *a, **kw):
            if timer.enabled and timer.match(geom):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                inner(geom, *a, **kw)
                e1.record()
                timer.events.append((e0, e1))
            else:
                inner(geom, *a, **kw)

        ops.conv_fwd = timed_conv_fwd

    def mean_ms(self):
        if not self.events:
            return None
        return sum(a.elapsed_time(b) for a, b in self.events) / len(self.events)


def cpu_baseline(args):
    """Time the CPU oracle (plain-PyTorch restatement of the reference step, pinned to the
    reference by tests/golden) on BASELINE.json configs[0]: B=4, K=2048, 3x32x128x128."""
    import torch.nn.functional as F
    from oracle import coclr_oracle as orc
    from model.pretrain import InfoNCE
    B, K = 4, 2048
    torch.set_num_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
    torch.manual_seed(0)
    model = InfoNCE(args.net, 128, K, 0.999, 0.07)
    sd = orc.training_state(model.state_dict())
    leaves = [sd[k] for k, _ in model.named_parameters() if sd[k].requires_grad]
    opt = torch.optim.Adam([{"params": p} for p in leaves], lr=1e-3, weight_decay=1e-5)
    times = []
    for step in range(1 + args.cpu_steps):
        g = torch.Generator().manual_seed(100 + step)
        block = torch.randn(B, 2, 3, args.seq_len, args.img_dim, args.img_dim, generator=g)
        perm = torch.randperm(B, generator=g)
        t0 = time.perf_counter()
        opt.zero_grad()
        (logits, labels), = orc.nce_step(sd, "infonce", args.net, [block], None, 128, K, 0.999,
                                         0.07, perm)
        F.cross_entropy(logits, labels).backward()
        opt.step()
        if step > 0:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": round(B / dt, 3), "unit": "clips/sec", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d timed steps (1 warm-up) of S3D InfoNCE K=2048 B=4 3x%dx%dx%d fwd+bwd+Adam, "
                      "%.2f s/step" % (len(times), args.seq_len, args.img_dim, args.img_dim, dt)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from coclr_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    from model.pretrain import InfoNCE, CoCLR

    K = args.moco_k or (2048 if world == 1 else 16384)
    B = args.batch
    torch.manual_seed(0)
    if args.model == "infonce":
        model = InfoNCE(args.net, 128, K, 0.999, 0.07)
    else:
        model = CoCLR(args.net, 128, K, 0.999, 0.07, topk=5)
        model.queue_label.fill_(1)       # queue "full": cross-modal mining active
        model.queue_vname.copy_(torch.randint(0, 2 ** 31, (K,)))
    model = model.cuda(local_rank)
    ddp = nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])
    params = [p for p in ddp.parameters() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5, fused=True)
    criterion = nn.CrossEntropyLoss().cuda(local_rank)
    ddp.train()
    if args.model == "coclr":
        model.sampler.eval()

    nblk = 2 if args.model == "coclr" else 1
    pool = [[synthetic_block(B, args.seq_len, args.img_dim, device, 1234 + rank + 1000 * i + 7 * j)
             for j in range(nblk)] for i in range(2)]
    vsrc = torch.randint(0, 2 ** 31, (B,), device=device)

    # dominant kernel: Conv_2c.conv1 = (1,3,3) 64->192 on (T/2, H/4, W/4), three launches per step
    # (q forward, k forward, q data-gradient of the same geometry class is 192->64 and excluded)
    tq, hq = args.seq_len // 2, args.img_dim // 4
    def is_dominant(g):
        return g.k == (1, 3, 3) and g.Cin == 64 and g.Cout == 192 and g.idim == (tq, hq, hq) \
            and g.d == (1, 1, 1)
    timer = KernelTimer(is_dominant)
    timer.install()

    def step(i):
        blocks = pool[i % 2]
        if args.model == "infonce":
            out, tgt = ddp(blocks[0])
            loss = criterion(out, tgt)
        else:
            out, mask = ddp(blocks[0], blocks[1], vsrc)
            loss = (- torch.log((torch.softmax(out, dim=1) * mask).sum(1))).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    for i in range(args.warmup):
        step(i)
    dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = True
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    t_host = time.perf_counter() - t0       # host enqueue time (no sync inside the loop)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.enabled = False
    tmax = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    final_loss = float(loss.detach())

    # the same dominant kernel alone on the chip (the in-step launches above share the CUs with the
    # key-encoder / weight-gradient streams): 20 back-to-back launches of Conv_2c.conv1
    iso_ms = None
    iso_algo = 0
    if rank == 0 and args.net == "s3d":
        from coclr_amd import ops, engine
        g = ops.conv_geom(B, 64, 192, (tq, hq, hq), (1, 3, 3), (1, 1, 1), (0, 1, 1))   # as the model
        run = engine.Run(device, save=False)
        xi = torch.randn(B, 64, tq, hq, hq, device=device)
        wi = torch.randn(192, 64, 1, 3, 3, device=device) * 0.05
        yi = torch.empty(B, 192, *g.odim, device=device)
        sti = torch.empty(2 * 192 * g.ntiles(), device=device)
        wpi = run.pack(wi, False, algo=g.algo)
        iso_algo = g.algo
        timer.enabled = False
        for _ in range(3):
            ops.conv_fwd(g, xi, wpi, yi, stats=sti)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.conv_fwd(g, xi, wpi, yi, stats=sti)
        e1.record()
        torch.cuda.synchronize()
        iso_ms = e0.elapsed_time(e1) / 20
        del xi, wi, yi, sti

    if rank == 0:
        ms = dt / args.steps * 1e3
        clips = B * world * args.steps / dt
        kms = timer.mean_ms()
        flops = 2.0 * B * 192 * 64 * 9 * tq * hq * hq          # algorithmic, per launch
        wino = bool(iso_ms) and iso_algo == 1
        kname = ("conv_wino_hw_kernel<8,6> Winograd F(2x2,3x3)" if wino
                 else "conv_igemm_kernel<1,3,3,8,64,128,4>")
        roof = None
        traffic = None
        tj = None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath) and args.net == "s3d" and B == 32:
            # HBM bytes per launch of this kernel from the PMC passes committed under profiles/
            # (counters cannot be read from inside the process)
            tj = json.load(open(tpath))
            if wino != ("wino" in tj["kernel"]):
                tj = None
        if tj is not None:
            traffic = {"bytes_per_launch": tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"],
                       "algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                       "source": tj["source"]}
        if kms:
            ach = flops / (kms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic,
                    "kernel": "%s (Conv_2c.conv1 64->192, %dx%dx%d, "
                              "N=%d; the query encoder's launches inside the timed steps, which share "
                              "the chip with the key-encoder stream)" % (kname, tq, hq, hq, B),
                    "launches_timed": len(timer.events), "avg_launch_ms": round(kms, 4),
                    "algorithmic_gflop_per_launch": round(flops / 1e9, 2)}
            if wino:
                # algorithmic = the direct convolution's FLOPs; the kernel issues 16/36 of them as MFMAs
                roof["mfma_gflop_per_launch"] = round(flops * 16.0 / 36.0 / 1e9, 2)
                roof["mfma_frac"] = round(ach * 16.0 / 36.0 / FP32_MFMA_PEAK_TFLOPS, 4)
            if iso_ms:
                # same kernel, same geometry, nothing else running: the kernel's own efficiency
                roof["isolated"] = {"avg_launch_ms": round(iso_ms, 4),
                                    "achieved": round(flops / (iso_ms * 1e-3) / 1e12, 2),
                                    "frac": round(flops / (iso_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
        # whole-step view against both rooflines (91.46 GF, 2145 MB per clip)
        step_view = {"tflops_per_gpu": round(91.46e9 * B / (ms * 1e-3) / 1e12, 2),
                     "frac_fp32_peak": round(91.46e9 * B / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                     "algorithmic_gbs_per_gpu": round(2145e6 * B / (ms * 1e-3) / 1e9, 1),
                     "frac_hbm_peak": round(2145e6 * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        rec = {
            "metric": "clips/sec (whole node) %s-%s seq_len=%d bs=%d/GPU" % (
                {"s3d": "S3D", "s3dg": "S3D-G", "r50": "R2D3D50"}.get(args.net, args.net),
                {"infonce": "InfoNCE", "coclr": "CoCLR"}[args.model], args.seq_len, B),
            "value": round(clips, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": {"workload": "%s %s moco-k=%d seq_len=%d img=%d bs=%d/GPU, DDP(nccl=RCCL) x%d, "
                                   "fwd+CE+bwd+Adam(fused, lr 1e-3, wd 1e-5)"
                                   % (args.net, args.model, K, args.seq_len, args.img_dim, B, world),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "excluded_caller_work": "accuracy meters/.item() syncs, dataloader+H2D",
                       "final_loss": round(final_loss, 4)},
            "roofline": roof, "step_roofline": step_view,
            "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # last thing on stdout (RCCL prints its banner lazily during the run)
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's banner sits in the C stdio buffer
        sys.stdout.flush()
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
