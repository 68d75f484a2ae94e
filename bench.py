#!/usr/bin/env python
"""Throughput benchmark of the CoCLR training hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's train_one_epoch body (main_nce.py:307-331)
on one synthetic batch already resident in HBM:
    logits, labels = DDP(InfoNCE('s3d'))(block)      q fwd, momentum update, shuffle-BN,
                                                     k fwd, logits, enqueue (+ collectives)
    loss = criterion(logits, labels); top1, top5 = calc_topk_accuracy(logits, labels, (1,5))
    optimizer.zero_grad(); loss.backward(); optimizer.step()
with the optimiser built exactly as main_nce.py:190-200 builds it: Adam(lr 1e-3, wd 1e-5) over ONE
PARAM GROUP PER TENSOR (470 groups).  Metric (BASELINE.json): clips/sec over the whole job =
B * world / step time, B = 32 clips of 3x32x128x128 per GPU, fp32.  Nothing is skipped inside the
timed region.

Legs (all in the one JSON line):
  value                    the drop-in: `torch.optim.Adam` resolved to the single-launch subclass by the
                           model.pretrain shim, loss + accuracy through coclr_amd.loss (device scalars)
  value_unmodified_caller  what main_nce.py gets WITHOUT editing a line of it: the shim's Adam (the script
                           constructs `optim.Adam`), but the script's own nn.CrossEntropyLoss,
                           utils.calc_topk_accuracy and the three `.item()` host syncs of
                           main_nce.py:314-327 per iteration
  value_split_stages       `value` with the backbone as one autograd node PER STAGE: the structure every
                           rank runs at world > 1 (DDP's all-reduce starts while early stages are
                           still in backward), i.e. the per-rank cost the scaling curve starts from
  value_caller_optimizer   the same step with torch's OWN Adam over the same 470 groups and
                           nn.CrossEntropyLoss (COCLR_PATCH_ADAM=0)
  roofline        dominant kernel (Conv_2c.conv1, spatial Winograd, 16-byte window DMA): MFMA FLOPs ACTUALLY ISSUED / time
                  / 157.3 TF, in-step (HIP events on the launch stream) and isolated; the
                  direct-convolution-equivalent figure is kept as `direct_equiv`
  roofline_hbm    largest BatchNorm+ReLU apply launch (Conv_1a.bn1): algorithmic bytes / time / 8 TB/s
  roofline_nce    the q.queue^T logits GEMM at K=16384: bytes and FLOPs / time, MFMA-busy from the
                  PMC pass committed under profiles/
  cpu_baseline    the CPU oracle (port of the reference step) on this host's cores, N=1 only
  self_check      one steady-state step run twice from the same snapshot -- as timed, and SERIAL (joins not
                  deferred, DDP's own bucket copies, all-gather exchange, no graph replay, one stream) --
                  with everything the step changes required to be bit-identical (bench_multi.py)
  value_k16384    N=1 only: the same step on moco-k=16384, the queue size every N > 1 run uses, so that
                  value(N) / value_k16384 is BASELINE config 3 over config 3

At world > 1 the ranks the driver starts are SUPERVISORS (bench_multi.supervise): each runs the measuring
process as a child; a child that hangs, dies or raises is replaced by one further down the degradation
ladder (bench_multi.RUNG_NAMES) and rank 0 still prints ONE valid JSON line, `multi_gpu.rung` naming what
had to be switched off and `multi_gpu.attempts` what was tried.
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
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip value_caller_optimizer and the isolated roofline micro-runs")
    ap.add_argument("--dry-run-host", action="store_true",
                    help="launch-contract rehearsal on the host (gloo, tiny shapes): only valid when "
                         "the caller has replaced coclr_amd.ops by the tests' ATen double "
                         "(tests/bench_dryrun.py); never a measurement")
    ap.add_argument("--no-self-check", action="store_true",
                    help="skip the fast-vs-serial bit-identity check of one step (bench_multi.SelfCheck)")
    ap.add_argument("--hang-timeout", type=float, default=150.0,
                    help="world > 1: seconds without progress (no new collective, no new step) after "
                         "which a rank reports the exchange it is stuck in and the job ends")
    ap.add_argument("--one-gpu-rehearsal", action="store_true",
                    help="world > 1 with EVERY rank on cuda:0 over gloo (RCCL refuses two ranks on one device): "
                         "the N > 1 bench path -- DDP, bucket hook, per-stage nodes, routed exchange, checked "
                         "step, instrumented collectives -- with the real kernels; only valid under "
                         "tests/bench_rehearse_gpu.py, which stages device payloads of the two collectives "
                         "gloo cannot carry through the host; never a measurement")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="time the CPU oracle step with --cpu-threads threads, print its record and exit (the child "
                         "process of the all-cores leg; no GPU is touched)")
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


def caller_topk_accuracy(output, target, topk=(1,)):
    """The launch script's accuracy helper (utils/utils.py:52-69), caller-side code in the reference's
    own formulation: top-k over the whole logits row, transpose, compare, slice sums."""
    maxk = max(topk)
    batch_size = target.size(0)
    _, pred = output.topk(maxk, 1, True, True)
    pred = pred.t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * (1 / batch_size) for k in topk]


class KernelTimer:
    """HIP-event timing of selected launches on the stream they are enqueued on (our kernels run
    on torch's current stream, so torch.cuda.Event brackets them)."""

    def __init__(self):
        self.events = {}
        self.enabled = False

    def wrap(self, module, name, key, match):
        inner = getattr(module, name)
        timer = self

        def timed(*a, **kw):
            from coclr_amd import plan as _plan
            rec = _plan.active()
            if rec is not None and match(*a, **kw):
                # the pass is being recorded into a launch plan: its replays never come through here again, so
                # the range of log entries this call makes is marked and LaunchPlan.replay brackets it
                # (plan.PROBE, installed below while the timer is enabled)
                b = len(rec.plan.entries)
                inner(*a, **kw)
                rec.plan.marks.append((b, len(rec.plan.entries), key))
            elif timer.enabled and match(*a, **kw):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                inner(*a, **kw)
                e1.record()
                timer.events.setdefault(key, []).append((e0, e1))
            else:
                inner(*a, **kw)

        setattr(module, name, timed)

    def enable(self, on):
        """Timing on / off -- for interpreted passes (the wrappers above) and for launch-plan replays."""
        from coclr_amd import plan as _plan
        self.enabled = bool(on)
        _plan.PROBE = (lambda key, e0, e1: self.events.setdefault(key, []).append((e0, e1))) if on else None

    def mean_ms(self, key):
        ev = self.events.get(key)
        if not ev:
            return None, 0
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)


class Watchdog:
    """world > 1: a first contact with RCCL / xGMI must end in evidence, not in the driver's timeout.
    A daemon thread watches (phase, number of collectives issued); when neither moves for `limit`
    seconds the rank writes which exchange it last issued (coclr_amd.parallel.LAST names the call site
    and the reference line it stands for) and ends the process; rank 0 also prints the contract's JSON
    line with "value": null and the diagnosis, so the driver's record says what hung."""

    def __init__(self, rank, world, limit, base_record):
        import threading
        from coclr_amd import parallel
        self.rank, self.world, self.limit, self.base = rank, world, limit, base_record
        self.parallel = parallel
        self.phase = ["start", 0]
        self.done = False
        self.parent = os.getppid()        # the supervisor (bench_multi.supervise): this process dies with it
        # set once the timed region has been measured: a hang in an OPTIONAL leg after it (host floor at
        # world > 1) must not cost the record -- the watchdog prints it and ends the job successfully
        self.complete_record = None       # rank 0: the record itself
        self.record_complete = False      # every rank: the timed region has been measured
        self.optional = False
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    def at(self, what, optional=False):
        self.phase[0], self.phase[1] = what, self.phase[1] + 1
        self.optional = optional

    def _run(self):
        import bench_multi
        seen, since = None, time.monotonic()
        while not self.done:
            time.sleep(min(1.0, self.limit / 4))
            if os.environ.get("COCLR_BENCH_CHILD") == "1" and os.getppid() != self.parent:
                os._exit(9)               # orphaned: a killed launcher must not leave a rank on a GPU
            cur = (self.phase[1], self.parallel.LAST[2])
            # what the supervisor reads if this process dies without a word (bench_multi.supervise)
            bench_multi.write_status(phase=self.phase[0], last_collective=self.parallel.LAST[0],
                                     collectives_issued=self.parallel.LAST[2])
            if cur != seen:
                seen, since = cur, time.monotonic()
                continue
            if time.monotonic() - since < self.limit:
                continue
            diag = {"error": "no progress for %.0f s" % self.limit, "rank": self.rank,
                    "phase": self.phase[0], "last_collective": self.parallel.LAST[0],
                    "collectives_issued": self.parallel.LAST[2]}
            print("bench watchdog: " + json.dumps(diag), file=sys.stderr, flush=True)
            try:
                import faulthandler
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)      # WHERE every thread is stuck
            except Exception:
                pass
            bench_multi.write_status(**diag)
            if self.optional and self.record_complete:
                if self.rank == 0 and self.complete_record is not None and not getattr(self, "printed", False):
                    rec = dict(self.complete_record, optional_leg_hung=diag)
                    print(json.dumps(rec), flush=True)
                os._exit(0)
            if self.rank == 0:
                rec = dict(self.base, value=None, ms_per_step=None, hang=diag)
                print(json.dumps(rec), flush=True)
            os._exit(5)


def cross_rank_digest(model, device):
    """What must be IDENTICAL on every rank after an optimiser step: the queue(s) and the pointer (every
    rank enqueues the same gathered keys, model/pretrain.py:82-96), the query parameters (same averaged
    gradients, same Adam) and the key parameters (same momentum update).  BatchNorm running statistics
    are rank-local until the next forward's broadcast and are left out.  Returns (names, float64 vector:
    plain and position-weighted sums, so that a permuted tensor does not pass)."""
    names, vals = [], []
    sd = model.state_dict()
    for k in sorted(sd):
        if k.startswith("queue"):
            t = sd[k].detach().double().reshape(-1)
            names.append(k)
            vals += [t.sum(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64, device=t.device)).sum()]
    for enc in ("encoder_q", "encoder_k"):
        tot = torch.zeros((), dtype=torch.float64, device=device)
        tot_abs = torch.zeros((), dtype=torch.float64, device=device)
        for p in getattr(model, enc).parameters():
            t = p.detach().double()
            tot += t.sum()
            tot_abs += t.abs().sum()
        names.append(enc + ".parameters")
        vals += [tot, tot_abs]
    for k in ("encoder_q.0.Conv_1a.conv1.weight", "encoder_q.0.Mixed_5c.branch1.1.conv2.weight",
              "encoder_q.0.conv1.weight", "encoder_q.4.weight", "encoder_k.4.weight"):
        if k in sd:
            t = sd[k].detach().double().reshape(-1)
            names.append(k)
            vals += [t.sum(), (t * torch.arange(1, t.numel() + 1, dtype=torch.float64, device=t.device)).sum()]
    return names, torch.stack([v.to(device) for v in vals])


def cross_rank_check(model, out, device, world):
    """One checked step's evidence: replicas bit-identical, logits finite on every rank."""
    names, digest = cross_rank_digest(model, device)
    got = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(got, digest)
    finite = torch.tensor([1.0 if bool(torch.isfinite(out).all()) else 0.0], device=device)
    dist.all_reduce(finite, op=dist.ReduceOp.MIN)
    worst, where = 0.0, None
    all_finite = bool(torch.isfinite(torch.stack(got)).all())
    identical = all_finite
    for r in range(1, world):
        if not torch.equal(got[r], got[0]):
            identical = False
            d = (got[r] - got[0]).abs()
            d = torch.where(torch.isfinite(d), d, torch.full_like(d, float("inf")))
            if where is None or float(d.max()) > worst:
                worst = float(d.max())
                where = "%s on rank %d" % (names[int(d.argmax()) // 2], r)
    if not all_finite and where is None:
        bad = (~torch.isfinite(torch.stack(got))).nonzero()[0]
        where = "%s on rank %d (not finite)" % (names[int(bad[1]) // 2], int(bad[0]))
    return {"replicas_identical": identical, "digests_finite": all_finite, "max_abs_digest_diff": worst,
            "first_mismatch": where,
            "fields": names, "logits_finite_on_every_rank": bool(finite.item() == 1.0),
            "what": "after one full step (fwd, loss, bwd, all-reduce, Adam, momentum, enqueue): float64 "
                    "digests (sum and position-weighted sum) of the queue(s), the pointer and the "
                    "query / key parameters, all-gathered and compared with rank 0's"}


def cpu_baseline(args, threads=None, steps=None):
    """Time the CPU oracle (plain-PyTorch restatement of the reference step, pinned to the
    reference by tests/golden) on BASELINE.json configs[0]: B=4, K=2048, 3x32x128x128."""
    import torch.nn.functional as F
    from oracle import coclr_oracle as orc
    from model.pretrain import InfoNCE
    from coclr_amd.optim import _TorchAdam
    B, K = 4, 2048
    torch.set_num_threads(max(1, min(threads or args.cpu_threads, os.cpu_count() or 1)))
    torch.manual_seed(0)
    model = InfoNCE(args.net, 128, K, 0.999, 0.07)
    sd = orc.training_state(model.state_dict())
    leaves = [sd[k] for k, _ in model.named_parameters() if sd[k].requires_grad]
    opt = _TorchAdam([{"params": p} for p in leaves], lr=1e-3, weight_decay=1e-5)
    times = []
    nsteps = steps or args.cpu_steps
    for step in range(1 + nsteps):
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
            "sample": "%d timed steps (1 warm-up) of S3D InfoNCE K=2048 B=4 3x%dx%dx%d fwd+bwd+Adam "
                      "through oracle/coclr_oracle.py (the reference's ATen CPU kernels) on %d of "
                      "this host's %d hardware threads, %.2f s/step; how the port compares with the "
                      "reference's OWN module on one host is measured by tools/cpu_port_vs_reference.py "
                      "(profiles/r05_cpu_port_vs_reference.txt) -- the reference itself does not exist on "
                      "the GPU box"
                      % (len(times), args.seq_len, args.img_dim, args.img_dim,
                         torch.get_num_threads(), os.cpu_count() or 0, dt)}


def cpu_baseline_all_cores(args, ncpu, fastest, budget_s=75.0):
    """The same oracle step with EVERY hardware thread of the host as ATen intra-op threads, in a child process
    with a hard time budget: on the 256-thread GPU host one oversubscribed step takes minutes (the first
    attempt at this leg ran past a 15-minute limit without finishing three), and the default bench run has to
    finish within a few.  Reports the rate, or that one warm-up + one timed step did not fit the budget."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-threads", str(ncpu),
           "--cpu-steps", "1", "--net", args.net, "--seq-len", str(args.seq_len), "--img-dim", str(args.img_dim)]
    t0 = time.perf_counter()
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s,
                             env=dict(os.environ, COCLR_QUIET="1"))
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        if out.returncode == 0 and line:
            r = json.loads(line[-1])
            return {"value": r["value"], "unit": r["unit"], "cores": r["cores"], "kind": r["kind"],
                    "sample": "the same oracle step, 1 timed step after 1 warm-up, on all %d hardware threads of "
                              "this host (child process, %.0f s); %d threads is the fastest setting measured: "
                              "cpu_baseline" % (ncpu, time.perf_counter() - t0, fastest["cores"])}
        why = "child exited with %d: %s" % (out.returncode, (out.stderr or "")[-200:])
    except subprocess.TimeoutExpired:
        why = "one warm-up + one timed step did not finish in %.0f s" % budget_s
    return {"value": None, "unit": "clips/sec", "cores": ncpu, "kind": "port",
            "sample": "the same oracle step on all %d hardware threads of this host: %s (oversubscribed ATen "
                      "convolutions; %d threads is the fastest setting measured, %.3f clips/s: cpu_baseline)"
                      % (ncpu, why, fastest["cores"], fastest["value"])}


def csrc_sha16():
    """Hash of the kernel sources of THIS build: counter files under profiles/ carry the hash of the build they
    were collected on (tools/traffic_json.py, tools/pmc_step_table.py) and are refused for any other."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "coclr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def counters_file(suffix):
    """(json, None) of the newest profiles/rNN_<suffix> collected on this build's kernels, or (None, why)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)), reverse=True)
    if not cands:
        return None, "no profiles/rNN_%s" % suffix
    tj = json.load(open(cands[0]))
    if tj.get("csrc_sha16") != csrc_sha16():
        return None, "%s was collected on other kernel sources (csrc %s, this build %s): re-run the PMC passes" % (
            os.path.basename(cands[0]), tj.get("csrc_sha16"), csrc_sha16())
    return tj, None


def nce_roofline(device, B):
    """The contrastive GEMM of BASELINE configs 3/5 (model/pretrain.py:175-182): q (B x 128) against
    the K=16384 queue, alone on the chip.  Bytes: queue + q + k + logits; FLOPs: 2*B*128*K."""
    import torch.nn.functional as F
    from coclr_amd import ops
    K, D, T = 16384, 128, 0.07
    g = torch.Generator(device=device).manual_seed(5)
    q = F.normalize(torch.randn(B, D, device=device, generator=g), dim=1)
    k = F.normalize(torch.randn(B, D, device=device, generator=g), dim=1)
    queue = F.normalize(torch.randn(D, K, device=device, generator=g), dim=0)
    logits = torch.empty(B, 1 + K, device=device)
    for _ in range(5):
        ops.nce_logits_fwd(q, k, queue, logits, T)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps):
        ops.nce_logits_fwd(q, k, queue, logits, T)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    flop = 2.0 * B * D * K
    byts = 4.0 * (D * K + 2 * B * D + B * (1 + K))
    rec = {"kernel": "nce_logits_fwd (l_pos + q.queue^T, /T) B=%d dim=128 K=16384, back-to-back "
                     "launches alone on the chip (includes the launch boundary)" % B,
           "avg_launch_us": round(us, 2), "bound": "hbm",
           "achieved": round(byts / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(byts / us / 1e3 / HBM_PEAK_GBS, 4),
           "algorithmic_mb_per_launch": round(byts / 1e6, 2),
           "tflops": round(flop / us / 1e6, 2),
           "mfma_frac_of_fp32_peak": round(flop / us / 1e6 / FP32_MFMA_PEAK_TFLOPS, 4)}
    for tag in ("r05", "r04", "r03", "r02"):
        pj = os.path.join(ROOT, "profiles", tag + "_nce_pmc.json")
        if os.path.exists(pj):
            rec["pmc"] = json.load(open(pj))
            break
    return rec


def dominant_kernel_name(algo):
    """The kernel the library selects for Conv_2c.conv1's forward at the bench shape (the selection of
    csrc/conv_igemm.hip `case 60`, restated: the choice depends on the geometry's algorithm and on two
    A/B switches read per call)."""
    if algo != 1:
        return "conv_igemm_kernel<1,3,3> direct implicit GEMM"
    x16 = os.environ.get("COCLR_WINO_X16", "1") != "0"
    w8 = os.environ.get("COCLR_WINO_W8", "1")[:1] != "0"
    if x16 and w8:
        return "conv_wino_hw8_kernel<8,3> Winograd F(2x2,3x3), two waves per SIMD, 16-byte window DMA"
    if x16:
        return "conv_wino_hw_kernel<8,3,true> Winograd F(2x2,3x3), one wave per SIMD, 16-byte window DMA"
    return "conv_wino_hw_kernel<8,6|10> Winograd F(2x2,3x3), 4-byte window DMA"


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: run the command the driver would have typed --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    <this script> <same arguments>` -- as a child, with its stdout / stderr passed through (rank 0's ONE JSON
    line included), and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + sys.argv[1:]
    print("bench: no launcher around --gpus %d, starting the ranks: %s" % (n, " ".join(cmd)), file=sys.stderr,
          flush=True)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL / hipIpc across processes need it here
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves, one per GPU, through the same launcher
        # the driver uses (this process only waits for it and hands its exit code on)
        sys.exit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run "
                         "--nproc-per-node %d, or unset WORLD_SIZE and let bench.py start the ranks)"
                         % (args.gpus, world, args.gpus))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1 and os.environ.get("COCLR_BENCH_CHILD") != "1" and \
            os.environ.get("COCLR_BENCH_LADDER", "1") != "0":
        # the ranks the driver starts supervise; the measuring processes are their children
        import bench_multi
        try:
            store = bench_multi._store()
        except Exception as e:
            # no store to agree over (a launcher without one): measure in this process, without the ladder
            print("bench: supervisors cannot reach the launcher's store (%s); running without the process "
                  "ladder" % e, file=sys.stderr, flush=True)
            store = None
        if store is not None:
            sys.exit(bench_multi.supervise(sys.argv[1:], args.hang_timeout, store))
    try:
        measure(args, world)
    except SystemExit:
        raise
    except BaseException as e:
        if world > 1:
            import traceback
            import bench_multi
            from coclr_amd import parallel as _par
            bench_multi.write_status(error=("%s: %s" % (type(e).__name__, e))[:600],
                                     last_collective=_par.LAST[0], collectives_issued=_par.LAST[2],
                                     phase=(_DOG[0].phase[0] if _DOG[0] is not None else "start"))
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(1)              # peers are waiting in a collective this rank will never enter
        raise


_DOG = [None]


def measure(args, world):
    if os.environ.get("COCLR_BENCH_TRACE"):
        # diagnosis: every N seconds, where is the host?  (stack of every thread to stderr)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["COCLR_BENCH_TRACE"]), repeat=True, file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run_host
    import datetime
    init = {"rank": rank, "world_size": world,
            "timeout": datetime.timedelta(seconds=max(60.0, 2 * args.hang_timeout))}
    if os.environ.get("COCLR_BENCH_INIT"):
        init["init_method"] = os.environ["COCLR_BENCH_INIT"]     # a child's own rendezvous (bench_multi)
    if dry:
        from coclr_amd import ops as _ops
        if _ops.conv_fwd.__module__ == "coclr_amd.ops":
            raise SystemExit("--dry-run-host needs the tests' double (python tests/bench_dryrun.py)")
        device = torch.device("cpu")
        dist.init_process_group("gloo", **init)
        torch.cuda.synchronize = lambda *a, **k: None
        args.no_extra_legs = args.no_cpu_baseline = True
    elif args.one_gpu_rehearsal:
        local_rank = 0
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
        dist.init_process_group("gloo", **init)
        args.no_extra_legs = args.no_cpu_baseline = True
    else:
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
        # RCCL's own watchdog fires after the bench's (which names the call site first)
        dist.init_process_group("nccl", device_id=device, **init)
        if world > 1:
            # the comparison legs (caller's optimiser, unmodified loop, isolated rooflines) are
            # single-GPU measurements; a scaling run reports `value` and nothing that could fail beside it
            args.no_extra_legs = True

    from coclr_amd import _lib
    _lib.load()                      # fail loudly if the HIP library is missing
    from model.pretrain import InfoNCE, CoCLR         # the shim also resolves torch.optim.Adam
    from coclr_amd import loss as L, ops, engine
    from coclr_amd import parallel as _par
    import coclr_amd.model.pretrain as _impl
    from coclr_amd.optim import Adam as NativeAdam, _TorchAdam
    import bench_multi
    if os.environ.get("COCLR_BENCH_FAULT"):
        bench_multi.inject_fault(os.environ["COCLR_BENCH_FAULT"], rank)      # tests only

    # N > 1: the K400 queue of BASELINE configs[2]; N = 1: configs[1] (the metric's own configuration) as
    # `value`, and the N > 1 queue as the extra leg `value_k16384`
    K = args.moco_k or (2048 if world == 1 else 16384)
    B = args.batch
    dog = None
    if world > 1:
        dog = _DOG[0] = Watchdog(rank, world, args.hang_timeout, {
            "metric": "clips/sec (whole node)", "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32", "data": "synthetic"})

    def build(K_):
        torch.manual_seed(0)
        if args.model == "infonce":
            m = InfoNCE(args.net, 128, K_, 0.999, 0.07)
        else:
            m = CoCLR(args.net, 128, K_, 0.999, 0.07, topk=5)
            m.queue_label.fill_(1)       # queue "full": cross-modal mining active
            m.queue_vname.copy_(torch.randint(0, 2 ** 31, (K_,)))
            m.queue_is_full = True
        return m if dry else m.cuda(local_rank)

    def wrap(m, hook=True):
        # main_nce.py:172; `hook=False` is the wrapper of the serial reference / of rungs >= 2
        saved = os.environ.get("COCLR_DDP_HOOK")
        if not hook:
            os.environ["COCLR_DDP_HOOK"] = "0"
        try:
            if dry:
                return nn.parallel.DistributedDataParallel(m)
            return nn.parallel.DistributedDataParallel(m, device_ids=[local_rank])
        finally:
            if not hook:
                if saved is None:
                    os.environ.pop("COCLR_DDP_HOOK", None)
                else:
                    os.environ["COCLR_DDP_HOOK"] = saved

    switches = bench_multi.Switches()
    rung = int(os.environ.get("COCLR_BENCH_RUNG", "0"))
    switches.apply(rung)
    model = build(K)
    wrappers = {}

    def make_ddp(hook):
        hook = bool(hook)
        if hook not in wrappers:
            # NOT `.train()` on the new wrapper: that would walk into the model and put CoCLR's frozen sampler
            # (eval mode, main_coclr.py:363) back into training mode -- which is how the self-check's first run
            # on CoCLR failed: the step before the second wrapper existed differed from every step after it
            wrappers[hook] = wrap(model, hook)
        return wrappers[hook]

    ddp = make_ddp(switches.wants_hook(rung))
    # main_nce.py:190-200: one param group per tensor, frozen ones included
    groups = [{"params": p} for _, p in ddp.named_parameters()]
    opt = torch.optim.Adam(groups, lr=1e-3, weight_decay=1e-5)
    assert isinstance(opt, NativeAdam), "the model.pretrain shim should have resolved torch.optim.Adam"
    criterion = L.CrossEntropyLoss()
    ddp.train()
    if args.model == "coclr":
        model.sampler.eval()

    nblk = 2 if args.model == "coclr" else 1
    pool = [[synthetic_block(B, args.seq_len, args.img_dim, device, 1234 + rank + 1000 * i + 7 * j)
             for j in range(nblk)] for i in range(2)]
    vsrc = torch.randint(0, 2 ** 31, (B,), device=device)

    # dominant kernel: Conv_2c.conv1 = (1,3,3) 64->192 on (T/2, H/4, W/4): the query encoder's forward
    # launches (the key encoder's replay from a hipGraph and cannot be bracketed)
    tq, hq = args.seq_len // 2, args.img_dim // 4
    th, hh = args.seq_len, args.img_dim // 2          # Conv_1a.bn1 output extent
    timer = KernelTimer()
    if not dry:
        timer.wrap(ops, "conv_fwd", "dominant",
                   lambda g, *a, **kw: g.k == (1, 3, 3) and g.Cin == 64 and g.Cout == 192
                   and g.idim == (tq, hq, hq) and g.d == (1, 1, 1))
        timer.wrap(ops, "bn_act_apply", "bn_apply",
                   lambda y, *a, **kw: tuple(y.shape) == (B, 64, th, hh, hh))

    acc = {}
    torch_ce = nn.CrossEntropyLoss().cuda(local_rank) if not dry else nn.CrossEntropyLoss()
    meters = {}
    live = {"ddp": ddp, "opt": opt, "pool": pool}

    def step(i, native=True, on=None):
        net = on if on is not None else live["ddp"]
        cur = live["opt"]
        blocks = live["pool"][i % 2]
        if args.model == "infonce":
            out, tgt = net(blocks[0])
            if native == "unmodified":
                # main_nce.py:313-327 verbatim: criterion, accuracy helper, three host reads
                loss = torch_ce(out, tgt)
                top1, top5 = caller_topk_accuracy(out, tgt, (1, 5))
                meters["top1"], meters["top5"], meters["loss"] = top1.item(), top5.item(), loss.item()
            elif native:
                loss = criterion(out, tgt)
                acc["top1"], acc["top5"] = L.calc_topk_accuracy(out, tgt, (1, 5))
            else:
                loss = nn.functional.cross_entropy(out, tgt)
        else:
            out, mask = net(blocks[0], blocks[1], vsrc[:blocks[0].shape[0]])
            if native:
                loss = L.multi_nce_loss(out, mask, drop_self=True)
                acc["top1"], acc["top5"] = L.calc_mask_accuracy(out, mask, (1, 5))
            else:
                loss = (- torch.log((torch.softmax(out, dim=1) * mask).sum(1))).mean()
        acc["logits"] = out.detach()
        cur.zero_grad(set_to_none=True)
        loss.backward()
        cur.step()
        return loss

    def timed_run(nsteps, native):
        assert _par.TIMINGS is None, "instrumented (serialising) collectives must be off in a timed region"
        dist.barrier()
        torch.cuda.synchronize()
        calls0 = _lib.CALLS[0]
        t0 = time.perf_counter()
        for i in range(nsteps):
            loss = step(i, native)
        t_host = time.perf_counter() - t0       # host enqueue time (no sync inside the loop)
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax), t_host, (_lib.CALLS[0] - calls0) / nsteps, float(loss.detach())

    def small_batch_floor(nf=12):
        """the same step on 4 clips per GPU, where the GPU is never the bottleneck: 4 + nf small steps in all
        put the queue pointer back on the big batch's grid (16 x 4 = 2 x 32 clips per rank)"""
        small = [[blk[:4].contiguous() for blk in blks] for blks in pool]
        live["pool"] = small
        try:
            done = 4
            for i in range(4):
                step(i)
            done += settle()                 # the small shape's launch plans are recorded outside the timed steps
            dtf, _, _, _ = timed_run(nf, True)
            done += nf
            while (done * 4) % B:            # ... and the pointer still ends on the big batch's grid
                step(0)
                done += 1
        finally:
            live["pool"] = pool
        step(0)                              # back to the benchmark shape (graphs are kept per shape)
        return round(dtf / nf * 1e3, 2)

    # ---- first contact and self-check -------------------------------------------------------------------
    multi = None
    self_check = None
    check = None
    # bucket views published (1), verified (2), joins deferred from (3): 4 is a steady-state step (on the host
    # double nothing is published or deferred -- no streams -- and a step costs seconds: 2)
    npre = 4 if not dry else 2
    if dog is not None:
        dog.at("first steps (buffer broadcast, gloo side group, peer mapping, exchange scheme chosen, "
               "graph capture, bucket views published and verified)")
    if world > 1 or not args.no_self_check:
        for i in range(npre):
            step(i)
        torch.cuda.synchronize()
    if world > 1:
        check = cross_rank_check(model, acc["logits"], device, world)
        if not check["replicas_identical"] or not check["logits_finite_on_every_rank"]:
            print("bench: cross-rank check FAILED: %s" % json.dumps(check), file=sys.stderr, flush=True)
    if not args.no_self_check:
        sc = bench_multi.SelfCheck(model, opt, make_ddp, lambda net, i: step(i, on=net), device, switches,
                                   torch.cuda.synchronize)
        self_check, rung = sc.run(rung, batch=npre, dog=dog)
        live["ddp"] = make_ddp(switches.wants_hook(rung))
        if not self_check["passed"]:
            print("bench: self-check FAILED on every rung: %s" % json.dumps(self_check["trials"]),
                  file=sys.stderr, flush=True)
        # the hooked wrapper publishes its bucket views again (the serial run took the engine's slots):
        # three steps until joins are deferred again, whatever --warmup says
        for i in range(3 if not dry else 1):
            step(i)
    if world > 1:
        dog.at("instrumented steps (collectives serialised and timed one by one)")
        _par.TIMINGS = []
        ninst = 2
        for i in range(ninst):
            step(i + 1)
        torch.cuda.synchronize()
        rows, _par.TIMINGS = _par.TIMINGS, None
        agg = {}
        for name, nbytes, ms in rows:
            a = agg.setdefault(name, [0, 0.0, nbytes])
            a[0] += 1
            a[1] += ms
        coll = [{"collective": name, "calls_per_step": round(a[0] / ninst, 2),
                 "ms_per_call": round(a[1] / a[0], 3), "mbytes": round(a[2] / 1e6, 3),
                 "gbs": round(a[2] / 1e6 / max(a[1] / a[0], 1e-6), 2)} for name, a in agg.items()]
        shuffle_ms = sum(a[1] / ninst for name, a in agg.items()
                         if "_shuffle" in name or "all_to_all" in name or "parked" in name
                         or ("all_gather_into_tensor" in name and a[2] > (1 << 20)))
        multi = {"rccl_ranks": world if not (dry or args.one_gpu_rehearsal) else 0,
                 "backend": dist.get_backend(),
                 "rung": rung, "rung_name": bench_multi.RUNG_NAMES[rung],
                 "rung_what": bench_multi.RUNG_WHAT[rung], "switches": switches.describe(),
                 "shuffle_mode": _impl._SHUFFLE_MODE, "shuffle_selection": dict(_impl._SHUFFLE_INFO),
                 "shuffle_exchange_ms_per_step_serialised": round(shuffle_ms, 3),
                 "split_stages": True, "cross_rank": check, "collectives": coll,
                 "collectives_ms_per_step_serialised": round(sum(a[1] for a in agg.values()) / ninst, 3),
                 "collectives_note": "each call bracketed by device synchronisations in two extra, untimed "
                                     "steps: its cost if nothing overlapped it, INCLUDING the wait for the "
                                     "slowest rank to arrive; the timed steps run them asynchronously"}
    def settle(native=True, limit=12):
        """Untimed steps until the launch plans of the structure in force have been recorded (engine.PLAN: four
        interpreted passes after the last signature change, then the recording pass) -- whatever --warmup says,
        a recording pass must not land in a timed region.  Returns the number of steps it took."""
        if dry or not engine.PLAN:
            return 0
        if world > 1:
            # every rank must run the SAME number of steps (each one is a set of collectives): no early exit on
            # this rank's own plan statistics -- warm-up (4) + recording + slack for a signature that settles late
            for n in range(8):
                step(n, native)
            return 8
        n = 0
        while n < limit:
            before = (engine.PLAN_STATS["recorded"], engine.PLAN_STATS["replayed"])
            step(n, native)
            n += 1
            if engine.PLAN_STATS["recorded"] == before[0] and engine.PLAN_STATS["replayed"] > before[1]:
                break                         # a step that only replayed
            if engine.PLAN_STATS["disabled"]:
                break
        return n

    if dog is not None:
        dog.at("warm-up steps")
    settled = settle()
    for i in range(args.warmup):
        step(i)
    if dog is not None:
        dog.at("timed steps")
    deferred0 = engine.DEFERRED[0]
    timer.enable(not dry)
    dt, t_host, calls_per_step, final_loss = timed_run(args.steps, True)
    timer.enable(False)
    deferred_per_step = (engine.DEFERRED[0] - deferred0) / max(1, args.steps)
    assert dry or live["opt"]._plan is not None, "the single-launch Adam did not run"
    # a number measured on a broken kernel is worse than no number
    if not (final_loss == final_loss and abs(final_loss) < 1e6):
        raise SystemExit("bench: the loss after the timed steps is %r -- refusing to report a throughput"
                         % final_loss)

    # ---- the same step as an unpatched caller gets it: torch's own Adam over the 470 groups ----------
    caller = None
    if not args.no_extra_legs:
        live["opt"] = _TorchAdam(groups, lr=1e-3, weight_decay=1e-5)
        for i in range(2):
            step(i, native=False)
        n2 = max(3, min(args.steps, 10))
        dt2, t_host2, _, _ = timed_run(n2, False)
        caller = {"value": round(B * world * n2 / dt2, 2), "ms_per_step": round(dt2 / n2 * 1e3, 3),
                  "steps": n2, "host_enqueue_ms_per_step": round(t_host2 / n2 * 1e3, 2),
                  "what": "torch.optim.Adam (torch's implementation) over %d single-tensor groups + "
                          "nn.functional.cross_entropy, everything else identical" % len(groups)}
        live["opt"] = opt

    # ---- the unmodified script's iteration, and the world>1 autograd structure --------------------------
    unmodified = split = None
    host_floor_split = None
    if not args.no_extra_legs and args.model == "infonce":
        for i in range(2):
            step(i, native="unmodified")
        settle("unmodified")
        n3 = max(3, min(args.steps, 10))
        dt3, t_host3, _, _ = timed_run(n3, "unmodified")
        unmodified = {"value": round(B * world * n3 / dt3, 2), "ms_per_step": round(dt3 / n3 * 1e3, 3),
                      "steps": n3,
                      "what": "THE NORTH STAR'S DROP-IN NUMBER: main_nce.py:307-331 exactly as written, not a "
                              "line of the caller edited -- the shim-resolved torch.optim.Adam, the script's "
                              "own nn.CrossEntropyLoss, utils.calc_topk_accuracy (ATen top-k over the "
                              "logits), top1.item() / top5.item() / loss.item() every iteration"}
        if args.net in ("s3d", "s3dg") and world == 1:
            from coclr_amd.backbone import s3dg as _s3dg
            saved_mode = _s3dg._SPLIT_MODE
            _s3dg._SPLIT_MODE = "1"
            try:
                for i in range(4):
                    step(i)
                settle()
                dt4, _, _, _ = timed_run(n3, True)
                if B >= 16:
                    host_floor_split = small_batch_floor()
            finally:
                _s3dg._SPLIT_MODE = saved_mode
            split = {"value": round(B * world * n3 / dt4, 2), "ms_per_step": round(dt4 / n3 * 1e3, 3),
                     "steps": n3, "host_floor_ms_per_step": host_floor_split,
                     "what": "`value` with one autograd node per backbone stage (COCLR_SPLIT_STAGES=1), "
                             "the structure used at world > 1; host_floor_ms_per_step = that structure on "
                             "4 clips per GPU (only the host paces it)"}
            step(0)

    # ---- host floor: the same step on 4 clips per GPU, where the GPU is never the bottleneck -----------
    host_floor = None
    if not args.no_extra_legs and args.model == "infonce" and B >= 16:
        host_floor = small_batch_floor()

    # ---- N = 1: the queue size of every N > 1 run, so that value(N) / value_k16384 compares like with like
    k16 = None
    if world == 1 and not args.no_extra_legs and K != 16384 and args.moco_k is None:
        model16 = build(16384)
        ddp16 = wrap(model16)
        ddp16.train()
        if args.model == "coclr":
            model16.sampler.eval()
        opt16 = torch.optim.Adam([{"params": p} for _, p in ddp16.named_parameters()], lr=1e-3,
                                 weight_decay=1e-5)
        live["ddp"], live["opt"] = ddp16, opt16
        for i in range(4):
            step(i)
        settle()
        n5 = max(5, min(args.steps, 20))
        dt5, _, _, _ = timed_run(n5, True)
        k16 = {"value": round(B * world * n5 / dt5, 2), "ms_per_step": round(dt5 / n5 * 1e3, 3), "steps": n5,
               "what": "the same step with moco-k=16384 (BASELINE configs[2]'s queue, what bench.py runs at "
                       "every N > 1): the N=1 point of the scaling curve on the SAME configuration"}
        live["ddp"], live["opt"] = make_ddp(switches.wants_hook(rung)), opt
        del model16, ddp16, opt16

    # ---- isolated micro-runs on rank 0: the dominant kernel, the largest BN apply, the NCE GEMM ------
    iso_ms = bn_iso_ms = None
    nce = None
    dom_algo = None
    if rank == 0 and args.net == "s3d" and not dry:
        dom_algo = ops.conv_geom(B, 64, 192, (tq, hq, hq), (1, 3, 3), (1, 1, 1), (0, 1, 1)).algo
    if rank == 0 and args.net == "s3d" and not args.no_extra_legs:
        g = ops.conv_geom(B, 64, 192, (tq, hq, hq), (1, 3, 3), (1, 1, 1), (0, 1, 1))   # as the model
        run = engine.Run(device, save=False)
        xi = torch.randn(B, 64, tq, hq, hq, device=device)
        wi = torch.randn(192, 64, 1, 3, 3, device=device) * 0.05
        yi = torch.empty(B, 192, *g.odim, device=device)
        sti = torch.empty(2 * 192 * g.ntiles(), device=device)
        wpi = run.pack(wi, False, algo=g.algo)
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
        yb = torch.randn(B, 64, th, hh, hh, device=device)
        zb = torch.empty_like(yb)
        sc_ = torch.rand(2, 64, device=device) + 0.5
        for _ in range(3):
            ops.bn_act_apply(yb, sc_[0], sc_[1], None, zb, True)
        e0.record()
        for _ in range(20):
            ops.bn_act_apply(yb, sc_[0], sc_[1], None, zb, True)
        e1.record()
        torch.cuda.synchronize()
        bn_iso_ms = e0.elapsed_time(e1) / 20
        del yb, zb
        nce = nce_roofline(device, B)

    rec = None
    if rank == 0:
        ms = dt / args.steps * 1e3
        clips = B * world * args.steps / dt
        kms, nk = timer.mean_ms("dominant")
        flops = 2.0 * B * 192 * 64 * 9 * tq * hq * hq          # direct-convolution FLOPs per launch
        wino = dom_algo == 1
        issued = flops * 16.0 / 36.0 if wino else flops         # F(2x2,3x3): 16 of 36 products
        roof = None
        traffic = None
        traffic_note = None
        if args.net == "s3d" and B == 32:
            # HBM bytes per launch of this kernel from the PMC passes committed under profiles/ (counters cannot
            # be read from inside the process) -- only when they were collected on THIS build's kernel sources
            tj, traffic_note = counters_file("traffic.json")
            if tj is not None and ("wino" in tj["kernel"]) == wino:
                traffic = {"bytes_per_launch": tj["fetch_bytes_per_launch"] + tj["write_bytes_per_launch"],
                           "algorithmic_bytes_per_launch": tj["algorithmic_bytes_per_launch"],
                           "mfma_busy_frac": tj.get("mfma_busy_frac"), "source": tj["source"]}
        if kms:
            ach = issued / (kms * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": "%s (Conv_2c.conv1 64->192, %dx%dx%d, N=%d; the query encoder's forward "
                              "launches inside the timed steps, sharing the chip with the key-encoder "
                              "stream)" % (dominant_kernel_name(dom_algo), tq, hq, hq, B),
                    "launches_timed": nk, "avg_launch_ms": round(kms, 4),
                    "mfma_gflop_per_launch": round(issued / 1e9, 2),
                    "what": "achieved = MFMA FLOPs the kernel actually issues (%s) / average launch time"
                            % ("16 of the 36 products of the direct convolution" if wino else
                               "the direct convolution's"),
                    "in_step_note": "HIP events on the launch stream bracket the launch: the in-step time includes "
                                    "waiting for CUs.  Since the query encoder is replayed from a launch plan the two "
                                    "encoders run in lockstep, and this persistent one-workgroup-per-CU kernel meets "
                                    "the key encoder's launch of the SAME kernel: the two cannot share a CU and run "
                                    "one after the other (in-step time ~ 2 x isolated; round 5's interpreted query "
                                    "pass lagged the key stream and measured 1.15 x).  `isolated` is the kernel",
                    "direct_equiv": {"algorithmic_gflop_per_launch": round(flops / 1e9, 2),
                                     "achieved": round(flops / (kms * 1e-3) / 1e12, 2),
                                     "frac": round(flops / (kms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}}
            if iso_ms:
                roof["isolated"] = {"avg_launch_ms": round(iso_ms, 4),
                                    "achieved": round(issued / (iso_ms * 1e-3) / 1e12, 2),
                                    "frac": round(issued / (iso_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                                    "direct_equiv_frac": round(flops / (iso_ms * 1e-3) / 1e12 /
                                                               FP32_MFMA_PEAK_TFLOPS, 4)}
        elif not dry:
            print("bench: the dominant kernel's launches were not timed (neither the interpreted wrappers nor the "
                  "launch plans' marks saw them): `roofline` is null", file=sys.stderr, flush=True)
        bms, nb = timer.mean_ms("bn_apply")
        roof_hbm = None
        if bms:
            byts = 2.0 * 4 * B * 64 * th * hh * hh            # read y, write z
            roof_hbm = {"bound": "hbm", "kernel": "bn_act_apply_kernel<relu> on Conv_1a.bn1 "
                                                  "(%dx64x%dx%dx%d): z = relu(y*scale+shift)" % (B, th, hh, hh),
                        "achieved": round(byts / (bms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(byts / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "algorithmic_mb_per_launch": round(byts / 1e6, 1), "launches_timed": nb,
                        "avg_launch_ms": round(bms, 4)}
            if bn_iso_ms:
                roof_hbm["isolated"] = {"avg_launch_ms": round(bn_iso_ms, 4),
                                        "achieved": round(byts / (bn_iso_ms * 1e-3) / 1e9, 1),
                                        "frac": round(byts / (bn_iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # whole-step view against both rooflines: algorithmic GFLOP / MB per training sample at the
        # reference clip size 3x32x128x128 (S3D InfoNCE 91.46 GF, 2145 MB; S3D CoCLR
        # 114.9 GF with the sampler's forward; r50 231.6 GF, 3758 MB); other shapes: not priced
        per = {("s3d", "infonce"): (91.46, 2145.0), ("s3d", "coclr"): (114.9, 2694.0),
               ("r50", "infonce"): (231.6, 3758.0)}.get((args.net, args.model))
        step_view = None
        if per is not None and args.seq_len == 32 and args.img_dim == 128:
            gf, mb = per
            pmc, pmc_note = counters_file("step_pmc.json") if (args.net, args.model, B) == ("s3d", "infonce", 32) \
                else (None, "counters are collected on the benchmark configuration only")
            step_view = {"mfma_issued_frac": None if pmc is None else round(
                             pmc["mfma_issued_gflop_per_step"] * 1e9 / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                         "mfma_issued_gflop_per_step": None if pmc is None else pmc["mfma_issued_gflop_per_step"],
                         "hbm_counter_gbs": None if pmc is None else round(
                             (pmc["hbm_fetch_mb_per_step"] + pmc["hbm_write_mb_per_step"]) * 1e6 / (ms * 1e-3) / 1e9, 1),
                         "counters": pmc_note or pmc["source"],
                         "what": "mfma_issued_frac = MFMA FLOPs the step's kernels ISSUE (SQ_INSTS_VALU_MFMA_MOPS_F32 "
                                 "x 512 summed over one step, Winograd forms counted as what they issue) / this "
                                 "run's step time / 157.3 TF; frac_fp32_peak below is in direct-convolution-"
                                 "equivalent FLOPs and flatters by the Winograd factor",
                         "gflop_per_sample": gf, "mb_per_sample": mb,
                         "tflops_per_gpu": round(gf * 1e9 * B / (ms * 1e-3) / 1e12, 2),
                         "frac_fp32_peak": round(gf * 1e9 * B / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                         "algorithmic_gbs_per_gpu": round(mb * 1e6 * B / (ms * 1e-3) / 1e9, 1),
                         "frac_hbm_peak": round(mb * 1e6 * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        rec = {
            "metric": "clips/sec (whole node) %s-%s seq_len=%d bs=%d/GPU" % (
                {"s3d": "S3D", "s3dg": "S3D-G", "r50": "R2D3D50"}.get(args.net, args.net),
                {"infonce": "InfoNCE", "coclr": "CoCLR"}[args.model], args.seq_len, B),
            "value": round(clips, 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "fp32",
            "data": "synthetic" if not (dry or args.one_gpu_rehearsal) else
                    ("DRY RUN ON THE HOST (ATen double, not a measurement)" if dry else
                     "REHEARSAL: every rank on ONE GPU over gloo (real kernels, not a measurement)"),
            "config": {"workload": "%s %s moco-k=%d seq_len=%d img=%d bs=%d/GPU, DDP(nccl=RCCL) x%d, "
                                   "fwd + loss + top-1/5 + bwd + Adam(lr 1e-3, wd 1e-5) over %d "
                                   "single-tensor param groups (main_nce.py:190-200); caller: "
                                   "import-swapped loss + accuracy (INTEGRATION.md section 3), the literally "
                                   "unmodified caller is value_unmodified_caller"
                                   % (args.net, args.model, K, args.seq_len, args.img_dim, B, world,
                                      len(groups)),
                       "global_batch": B * world, "parallelism": "dp%d" % world,
                       "optimizer": "coclr_amd.optim.Adam (torch.optim.Adam resolved by the "
                                    "model.pretrain shim): one launch per step",
                       "loss": "coclr_amd.loss (loss + top-1/top-5 in one pass, device scalars)",
                       "ddp": "DistributedDataParallel(model, device_ids=[gpu]) as main_nce.py:172; "
                              "gradient_as_bucket_view=%s (the shim's default for this model, "
                              "COCLR_PATCH_DDP=0 restores torch's False: about -3 %%)"
                              % getattr(live["ddp"], "gradient_as_bucket_view", None),
                       "scaling_note": "N > 1 runs moco-k=16384 (BASELINE configs[2]); the N=1 point on that "
                                       "queue size is value_k16384 of the N=1 line",
                       "excluded_caller_work": "meters' .item() syncs, dataloader + H2D",
                       "final_loss": round(final_loss, 4)},
            "roofline": roof, "roofline_hbm": roof_hbm, "roofline_nce": nce,
            "step_roofline": step_view,
            "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 2),
            "host_floor_ms_per_step": host_floor,
            "host_note": "host_enqueue includes the time the host sits in a launch because the hardware "
                         "queue is full (no synchronising call inside a step, tools/find_syncs.py); "
                         "host_floor = the same step at 4 clips/GPU, where only the host paces it",
            "abi_calls_per_step": round(calls_per_step, 1),
            "launch_plans": {"on": bool(engine.PLAN), "recorded": engine.PLAN_STATS["recorded"],
                             "replayed_passes": engine.PLAN_STATS["replayed"],
                             "disabled": engine.PLAN_STATS["disabled"][:3], "settle_steps": settled,
                             "what": "the query encoder's C-ABI calls re-issued from a recorded log with "
                                     "pre-marshalled arguments (coclr_amd/plan.py): same launches on the same "
                                     "streams, ~2 us of host time per call instead of ~30"},
            "deferred_joins_per_step": round(deferred_per_step, 2),
            "self_check": self_check,
        }
        if self_check is not None and not self_check["passed"]:
            rec["self_check_failed"] = True
        if check is not None and not (check["replicas_identical"] and check["logits_finite_on_every_rank"]):
            rec["cross_rank_failed"] = True
        if multi is not None:
            rec["multi_gpu"] = multi
        if unmodified is not None:
            rec["value_unmodified_caller"] = unmodified
        if split is not None:
            rec["value_split_stages"] = split
        if caller is not None:
            rec["value_caller_optimizer"] = caller
        if k16 is not None:
            rec["value_k16384"] = k16
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(args)
            # SURVEY 8d asks for the reference's CPU path on the box's host cores: the same oracle step on EVERY
            # hardware thread the host has (slower than 16 on the 256-thread GPU host: oversubscribed ATen
            # convolutions), beside the fastest thread count above
            ncpu = os.cpu_count() or 1
            if ncpu > rec["cpu_baseline"]["cores"]:
                rec["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args, ncpu, rec["cpu_baseline"])

    # ---- N > 1: host floor of the per-stage structure; optional -- a hang here must not cost the record --
    if world > 1 and args.model == "infonce" and B >= 16 and K % (4 * world) == 0:
        dog.complete_record, dog.record_complete = rec, True
        dog.at("host floor (4 clips per GPU) -- optional leg", optional=True)
        if bench_multi.FLOOR_HANG:
            time.sleep(3600)             # tests: a rank lost in the optional leg
        try:
            floor = small_batch_floor()
            if rank == 0:
                rec["multi_gpu"]["host_floor_ms_per_step"] = floor
        except Exception as e:          # the record is complete without it
            if rank == 0:
                rec["multi_gpu"]["host_floor_error"] = ("%s: %s" % (type(e).__name__, e))[:300]
    if dog is not None:
        dog.at("final barrier", optional=True)
    dist.barrier()

    def emit():
        if rank == 0:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL's banner sits in the C stdio buffer
            sys.stdout.flush()
            print(json.dumps(rec), flush=True)

    if dog is not None:
        # world > 1 (a supervised child): the line goes out BEFORE the process group is torn down -- a
        # destroy_process_group() that never returns must not cost a finished measurement (the supervisor takes
        # the last JSON line whatever follows it on stdout; the watchdog ends a stuck teardown with exit code 0)
        emit()
        dog.complete_record, dog.record_complete, dog.printed = None, True, True
        dog.at("destroy_process_group", optional=True)
        bench_multi.write_status(phase="done")
        dist.destroy_process_group()
        dog.done = True
        return
    bench_multi.write_status(phase="done")
    dist.destroy_process_group()
    emit()          # world 1: last thing on stdout (RCCL prints its banner lazily during the run)


if __name__ == "__main__":
    main()
