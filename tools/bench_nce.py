"""Contrastive-head kernels at the benchmark shapes (B=32, dim=128, K in {2048, 16384}): the fused
[l_pos | q.queue]/T logits (model/pretrain.py:175-182), its backward, the CoCLR similarity GEMM +
top-k positive mask (:405-413).  GPU box only; `tools/pmc_layers.sh` runs it under rocprofv3 --pmc
(PMC_SCRIPT=tools/bench_nce.py)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from coclr_amd import ops

dev = torch.device("cuda")
B, D, T = int(os.environ.get("B", "32")), 128, 0.07
REPS = int(os.environ.get("REPS", "50"))


def timeit(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us


print("%-8s | %10s %8s %8s | %10s | %10s %10s" % ("K", "logits us", "GFLOP/s", "GB/s", "bwd us",
                                                 "sim us", "mask us"))
for K in [int(v) for v in os.environ.get("KS", "2048,16384").split(",")]:
    torch.manual_seed(0)
    q = F.normalize(torch.randn(B, D, device=dev), dim=1)
    k = F.normalize(torch.randn(B, D, device=dev), dim=1)
    queue = F.normalize(torch.randn(D, K, device=dev), dim=0)
    logits = torch.empty(B, 1 + K, device=dev)
    t_f = timeit(lambda: ops.nce_logits_fwd(q, k, queue, logits, T))
    dl = torch.randn_like(logits)
    dq = torch.empty(B, D, device=dev)
    splits = max(1, min(K // 128, 256))
    ws = torch.empty(max(1, ops.gemm_workspace(B, D, K, splits)), device=dev)
    t_b = timeit(lambda: ops.nce_logits_bwd(dl, k, queue, dq, ws, T, splits))
    sim = torch.empty(B, K, device=dev)
    t_s = timeit(lambda: ops.gemm(q, D, 1, queue, K, 1, sim, K, None, B, K, D))
    src = torch.randint(0, 1000, (B,), device=dev)
    names = torch.randint(0, 1000, (K,), device=dev)
    mask = torch.empty(B, 1 + K, dtype=torch.uint8, device=dev)
    t_m = timeit(lambda: ops.positive_mask(sim, src, names, mask, 5))
    flop = 2.0 * B * D * K
    byts = 4.0 * (D * K + B * D * 2 + B * (1 + K))
    print("%-8d | %10.2f %8.0f %8.0f | %10.2f | %10.2f %10.2f" % (
        K, t_f, flop / t_f / 1e3, byts / t_f / 1e3, t_b, t_s, t_m))
