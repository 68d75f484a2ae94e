"""Per-launch times of the projection head's products, alone on the chip (HIP events around 200 back-to-back
launches): coclr_gemm (product + fold) followed by the row operation as its own launch, against
coclr_gemm_fused (the fold kernel applies the row operation) at several split counts.
(profiles/r05_head_probe_lastblock.txt is this probe on the FIRST form of coclr_gemm_fused, where the last
workgroup of a tile folded the partials inside the product's launch: 2-8x slower -- a device-scope fence per
workgroup writes back / invalidates L2 across the eight XCDs, and one workgroup folds a whole tile.)  B = 32, Cf = Ch = 1024, D = 128, S = 64 (S3D) and K = 2048 / 16384."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coclr_amd import ops

dev = torch.device("cuda", 0)
B, C, D, S = 32, 1024, 128, 64
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)


def timeit(fn, reps=200):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


h0, w1, b1, h1 = R(B, C), R(C, C) * 0.03, R(C), torch.empty(B, C, device=dev)
w2, b2 = R(D, C) * 0.03, R(D)
q, inv = torch.empty(B, D, device=dev), torch.empty(B, device=dev)
f = torch.empty(B, D, device=dev)


def two_launch(M, N, K, splits, a, sa, b_, sb, c, ldc, bias=None, relu=False, alpha=1.0):
    ws = torch.empty(max(1, ops.gemm_workspace(M, N, K, splits)), device=dev)
    return lambda: ops.gemm(a, sa[0], sa[1], b_, sb[0], sb[1], c, ldc, bias, M, N, K, alpha=alpha, relu=relu,
                            splits=splits, workspace=ws)


def fused(M, N, K, splits, a, sa, b_, sb, c, ldc, bias=None, relu=False, alpha=1.0, **kw):
    ws = torch.empty(ops.gemm_fused_workspace(M, N, K, splits), device=dev)
    return lambda: ops.gemm_fused(a, sa[0], sa[1], b_, sb[0], sb[1], c, ldc, bias, M, N, K, alpha=alpha,
                                  relu=relu, splits=splits, workspace=ws, **kw)


print("fc1 forward (32x1024x1024, + bias + ReLU)")
print("  gemm+reduce splits=32: %.1f us   relu_fwd: %.1f us" % (
    timeit(two_launch(B, C, C, 32, h0, (C, 1), w1, (1, C), h1, C, b1)), timeit(lambda: ops.relu_fwd(h1, h1))))
for sp in (4, 8, 16, 32):
    print("  fused splits=%d: %.1f us" % (sp, timeit(fused(B, C, C, sp, h0, (C, 1), w1, (1, C), h1, C, b1, relu=True))))
print("fc2 forward (32x128x1024, + bias) + F.normalize")
print("  gemm+reduce splits=32: %.1f us   l2norm_fwd: %.1f us" % (
    timeit(two_launch(B, D, C, 32, h1, (C, 1), w2, (1, C), f, D, b2)), timeit(lambda: ops.l2norm_fwd(f, q, inv))))
for sp in (2, 4, 8, 16, 32):
    print("  fused splits=%d: %.1f us" % (sp, timeit(fused(B, D, C, sp, h1, (C, 1), w2, (1, C), q, D, b2, mode=2,
                                                           out2=inv, f=1e-12))))
for K in (2048, 16384):
    dl, queue, k = R(B, 1 + K), R(D, K), R(B, D)
    dq, df = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
    sp0 = max(1, min(K // 128, 256))
    ws = torch.empty(ops.gemm_workspace(B, D, K, sp0), device=dev)
    print("logits backward K=%d" % K)
    print("  nce_logits_bwd (gemm+reduce+lpos) splits=%d: %.1f us   l2norm_bwd: %.1f us" % (
        sp0, timeit(lambda: ops.nce_logits_bwd(dl, k, queue, dq, ws, 0.07, sp0)),
        timeit(lambda: ops.l2norm_bwd(dq, q, inv, df))))
    for sp in (8, 16, 32, 64, 128):
        if sp > K // 32:
            continue
        print("  fused splits=%d: %.1f us" % (sp, timeit(fused(
            B, D, K, sp, dl[:, 1:], (1 + K, 1), queue, (1, K), df, D, alpha=1 / 0.07, mode=3, ep_a=dl, lda=1 + K,
            ep_b=k, ep_y=q, inv_norm=inv, f=1 / 0.07))))
df = R(B, D)
dh1 = torch.empty(B, C, device=dev)
print("fc2 backward d(hidden) (32x1024x128) + ReLU backward")
print("  gemm+reduce splits=4: %.1f us   relu_bwd: %.1f us" % (
    timeit(two_launch(B, C, D, 4, df, (D, 1), w2, (C, 1), dh1, C)), timeit(lambda: ops.relu_bwd(dh1, h1, dh1))))
for sp in (1, 2, 4):
    print("  fused splits=%d: %.1f us" % (sp, timeit(fused(B, C, D, sp, df, (D, 1), w2, (C, 1), dh1, C, mode=1,
                                                           ep_a=h1, lda=C))))
dw2, db2 = torch.empty(D, C, device=dev), torch.empty(D, device=dev)
print("fc2 weight gradient (128x1024x32) + bias gradient")
print("  gemm: %.1f us   colsum: %.1f us" % (timeit(two_launch(D, C, B, 1, df, (1, D), h1, (C, 1), dw2, C)),
                                            timeit(lambda: ops.colsum(df, db2))))
print("  fused (rowsum): %.1f us" % timeit(lambda: ops.gemm_fused(df, 1, D, h1, C, 1, dw2, C, None, D, C, B,
                                                                 rowsum=db2)))
dh0 = torch.empty(B, C, device=dev)
dx = torch.empty(B, C, 4, 4, 4, device=dev)
print("fc1 backward d(pooled) (32x1024x1024) + average-pool backward (8 MB)")
print("  gemm+reduce splits=32: %.1f us   avgpool_bwd: %.1f us" % (
    timeit(two_launch(B, C, C, 32, dh1, (C, 1), w1, (C, 1), dh0, C)),
    timeit(lambda: ops.global_avgpool_bwd(dh0.view(B, C, 1, 1, 1), dx))))
for sp in (4, 8, 16, 32):
    print("  fused (mode 4) splits=%d: %.1f us" % (sp, timeit(fused(B, C, C, sp, dh1, (C, 1), w1, (C, 1), dx, 0,
                                                                     mode=4, S=S))))
    print("  fused (mode 0) splits=%d: %.1f us" % (sp, timeit(fused(B, C, C, sp, dh1, (C, 1), w1, (C, 1), dh0, C))))
dw1, db1 = torch.empty(C, C, device=dev), torch.empty(C, device=dev)
print("fc1 weight gradient (1024x1024x32) + bias gradient")
print("  gemm: %.1f us   colsum: %.1f us" % (timeit(two_launch(C, C, B, 1, dh1, (1, C), h0, (C, 1), dw1, C)),
                                            timeit(lambda: ops.colsum(dh1, db1))))
print("  fused (rowsum): %.1f us" % timeit(lambda: ops.gemm_fused(dh1, 1, C, h0, C, 1, dw1, C, None, C, C, B,
                                                                 rowsum=db1)))
x = R(B, C, 4, 4, 4)
print("avgpool_fwd: %.1f us" % timeit(lambda: ops.global_avgpool_fwd(x, h0.view(B, C, 1, 1, 1))))
