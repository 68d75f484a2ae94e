// MoCo / InfoNCE / CoCLR head kernels for gfx950 (model/pretrain.py):
//   * skinny fp32-MFMA GEMM with arbitrary operand strides, split-K and a fused
//     epilogue: the q . queue^T contraction (pretrain.py:176), its backward, the
//     CoCLR cross-modal similarity (pretrain.py:405) and the 1x1x1 projection
//     convs on pooled features (pretrain.py:52,54)
//   * L2 normalise fwd/bwd (pretrain.py:154,167,380), l_pos (pretrain.py:175)
//   * (the multi-tensor momentum update, pretrain.py:76-80, lives in optim.hip)
//   * FIFO queue enqueue with a device-resident pointer (pretrain.py:82-96,321-341)
//   * per-row top-k mining + positive-mask build (pretrain.py:397-413, 267-269)
//   * row gather for shuffle-BN (pretrain.py:124,143), ReLU, column sums
#include "common.h"
#include "../../include/coclr_hip.h"
#include <math.h>

namespace {

// The epilogue arithmetic of the head, with every rounding pinned: the same value must come out whether an
// operation runs as its own launch (l2norm_fwd_kernel after the fold) or inside the fold kernel of
// coclr_gemm_fused -- the compiler is otherwise free to contract `a * b + c` into one fused multiply-add in
// one kernel and not in the other (it did: F.normalize differed in the last bit between the two forms).
__device__ __forceinline__ float ep_scale_bias(float s, float alpha, const float* bias, int n) {
  float v = __fmul_rn(s, alpha);
  if (bias) v = __fadd_rn(v, bias[n]);
  return v;
}
__device__ __forceinline__ float ep_fma(float a, float b, float c) { return __fmaf_rn(a, b, c); }
__device__ __forceinline__ float ep_inv_norm(float ss, float eps) { return __fdiv_rn(1.f, fmaxf(__fsqrt_rn(ss), eps)); }
// (dy - y * dot) * inv
__device__ __forceinline__ float ep_norm_bwd(float dy, float y, float dot, float inv) {
  return __fmul_rn(__fmaf_rn(-y, dot, dy), inv);
}
// dq + (dl0 * inv_T) * k
__device__ __forceinline__ float ep_lpos(float dq, float dl0, float inv_T, float k) {
  return __fmaf_rn(__fmul_rn(dl0, inv_T), k, dq);
}

// ---------------------------------------------------------------------------
// C[m][n] (+)= act(alpha * sum_k A(m,k) B(k,n) + bias[n])
// A(m,k) at a[m*sam + k*sak], B(k,n) at b[k*sbk + n*sbn].
// TA: A is m-contiguous (sam == 1) else k-contiguous; TB: B is k-contiguous
// (sbk == 1) else n-contiguous.  Tile 32 x 128 x 32, 4 waves, one
// v_mfma_f32_32x32x2_f32 column block per wave.
// ---------------------------------------------------------------------------
struct GemmArgs {
  const float* a; const float* b; float* c; const float* bias;
  long sam, sak, sbk, sbn, ldc;
  int M, N, K, kslice;
  float alpha;
  int relu, accumulate, splits;
  float* part;   // [splits][M][N] when the product goes through the fold kernel
  // ---- coclr_gemm_fused: the row-level operation that follows the product, applied by the kernel that folds
  // the split-K partials (one pass over the partials instead of fold + one or two more launches)
  int to_part, ep_mode, ep_S;
  const float* ep_a; long ep_lda;
  const float* ep_b; const float* ep_y; const float* ep_inv;
  float* ep_out2; float ep_f;
  float* rowsum;   // splits == 1: rowsum[m] = sum_k A(m,k), written by the workgroups of the first tile column
};

template <bool TA, bool TB>
__global__ void __launch_bounds__(256)
gemm32x128_kernel(const GemmArgs g) {
  constexpr int BK = 32, LDA = 33, LDB = 129;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 32;
  const int split = blockIdx.z;
  const int kbeg = split * g.kslice;
  int kend = kbeg + g.kslice;
  if (kend > g.K) kend = g.K;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float rs = 0.f;

  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int m, k;
      if (TA) { m = e & 31; k = e >> 5; } else { k = e & 31; m = e >> 5; }
      const int gm = m0 + m, gk = k0 + k;
      As[k * LDA + m] = (gm < g.M && gk < kend) ? g.a[gm * g.sam + gk * g.sak] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + i * 256;
      int n, k;
      if (TB) { k = e & 31; n = e >> 5; } else { n = e & 127; k = e >> 7; }
      const int gn = n0 + n, gk = k0 + k;
      Bs[k * LDB + n] = (gn < g.N && gk < kend) ? g.b[gk * g.sbk + gn * g.sbn] : 0.f;
    }
    __syncthreads();
    if (g.rowsum && blockIdx.x == 0 && tid < 32) {
#pragma unroll
      for (int k = 0; k < BK; ++k) rs += As[k * LDA + tid];
    }
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const int k = 2 * s + half;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[k * LDA + l31], Bs[k * LDB + wave * 32 + l31],
                                                 acc, 0, 0, 0);
    }
  }

  if (g.rowsum && blockIdx.x == 0 && tid < 32 && m0 + tid < g.M) g.rowsum[m0 + tid] = rs;
  const int gn = n0 + wave * 32 + l31;
  if (gn >= g.N) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int gm = m0 + (i & 3) + 8 * (i >> 2) + 4 * half;
    if (gm >= g.M) continue;
    if (g.to_part) {
      g.part[((long)split * g.M + gm) * g.N + gn] = acc[i];
    } else {
      float v = ep_scale_bias(acc[i], g.alpha, g.bias, gn);
      if (g.relu) v = fmaxf(v, 0.f);
      float* d = g.c + gm * g.ldc + gn;
      *d = g.accumulate ? __fadd_rn(*d, v) : v;
    }
  }
}

__global__ void gemm_splitk_reduce_kernel(const GemmArgs g) {
  const long MN = (long)g.M * g.N;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < MN;
       e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < g.splits; ++k) s += g.part[(long)k * MN + e];
    const long m = e / g.N;
    const int n = (int)(e - m * g.N);
    float v = ep_scale_bias(s, g.alpha, g.bias, n);
    if (g.relu) v = fmaxf(v, 0.f);
    if (g.ep_mode == 1) v = g.ep_a[m * g.ep_lda + n] > 0.f ? v : 0.f;          // ReLU backward
    float* d = g.c + m * g.ldc + n;
    *d = g.accumulate ? __fadd_rn(*d, v) : v;
  }
}

// fold + global-average-pool backward: output element (m, n, s) = fold(m, n) / S.  The S threads of a
// plane read the same partials (one transaction per wave and split) and write S consecutive floats.
__global__ void gemm_splitk_reduce_expand_kernel(const GemmArgs g) {
  const long MN = (long)g.M * g.N;
  const long total = MN * g.ep_S;
  const float inv = 1.f / (float)g.ep_S;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long mn = e / g.ep_S;
    float s = 0.f;
    for (int k = 0; k < g.splits; ++k) s += g.part[(long)k * MN + mn];
    const float v = ep_scale_bias(s, g.alpha, g.bias, (int)(mn % g.N));
    g.c[e] = __fmul_rn(v, inv);
  }
}

// fold + a ROW operation, one wave per output row (lane l owns columns l, l + 64, ...; N <= 64 * RC):
//   mode 2  F.normalize: c = v / max(||v||, f), out2 = 1 / max(||v||, f)        (as l2norm_fwd_kernel)
//   mode 3  v += a[m*lda] * f * b[m][n] (l_pos term), then c = (v - y <y, v>) * inv_norm[m]
//                                                                                   (as l2norm_bwd_kernel)
template <int RC>
__global__ void __launch_bounds__(256)
gemm_splitk_reduce_rows_kernel(const GemmArgs g) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= g.M) return;
  const long MN = (long)g.M * g.N;
  float v[RC];
#pragma unroll
  for (int j = 0; j < RC; ++j) {
    const int n = lane + 64 * j;
    float s = 0.f;
    if (n < g.N)
      for (int k = 0; k < g.splits; ++k) s += g.part[(long)k * MN + (long)row * g.N + n];
    float x = 0.f;
    if (n < g.N) {
      x = ep_scale_bias(s, g.alpha, g.bias, n);
      if (g.relu) x = fmaxf(x, 0.f);
      if (g.ep_mode == 3) x = ep_lpos(x, g.ep_a[row * g.ep_lda], g.ep_f, g.ep_b[(long)row * g.N + n]);
    }
    v[j] = x;
  }
  if (g.ep_mode == 2) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < RC; ++j)
      if (lane + 64 * j < g.N) ss = ep_fma(v[j], v[j], ss);
    ss = wave_sum(ss);
    const float inv = ep_inv_norm(ss, g.ep_f);
#pragma unroll
    for (int j = 0; j < RC; ++j)
      if (lane + 64 * j < g.N) g.c[row * g.ldc + lane + 64 * j] = __fmul_rn(v[j], inv);
    if (lane == 0 && g.ep_out2) g.ep_out2[row] = inv;
  } else {
    float y[RC];
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < RC; ++j) {
      y[j] = lane + 64 * j < g.N ? g.ep_y[(long)row * g.N + lane + 64 * j] : 0.f;
      if (lane + 64 * j < g.N) dot = ep_fma(v[j], y[j], dot);
    }
    dot = wave_sum(dot);
    const float inv = g.ep_inv[row];
#pragma unroll
    for (int j = 0; j < RC; ++j)
      if (lane + 64 * j < g.N) g.c[row * g.ldc + lane + 64 * j] = ep_norm_bwd(v[j], y[j], dot, inv);
  }
}

// ---------------------------------------------------------------------------
// rows of length D: y = x / max(||x||, eps); one wave per row
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ inv_norm,
                  int rows, int D, float eps) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xp = x + (long)row * D;
  float ss = 0.f;
  for (int i = lane; i < D; i += 64) ss = ep_fma(xp[i], xp[i], ss);
  ss = wave_sum(ss);
  const float inv = ep_inv_norm(ss, eps);
  for (int i = lane; i < D; i += 64) y[(long)row * D + i] = __fmul_rn(xp[i], inv);
  if (lane == 0 && inv_norm) inv_norm[row] = inv;
}

// dx = (dy - y * <y, dy>) * inv_norm
__global__ void __launch_bounds__(256)
l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                  const float* __restrict__ inv_norm, float* __restrict__ dx, int rows, int D) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* dyp = dy + (long)row * D;
  const float* yp = y + (long)row * D;
  float dot = 0.f;
  for (int i = lane; i < D; i += 64) dot = ep_fma(dyp[i], yp[i], dot);
  dot = wave_sum(dot);
  const float inv = inv_norm[row];
  for (int i = lane; i < D; i += 64) dx[(long)row * D + i] = ep_norm_bwd(dyp[i], yp[i], dot, inv);
}

// logits[b][0] = <q_b, k_b> * inv_T
__global__ void __launch_bounds__(256)
lpos_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, float* logits, int rows,
                int D, long ldl, float inv_T) {
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += q[(long)row * D + i] * k[(long)row * D + i];
  s = wave_sum(s);
  if (lane == 0) logits[row * ldl] = s * inv_T;
}

// ---------------------------------------------------------------------------
// logits[b][0] = <q_b, k_b>/T,  logits[b][1+j] = <q_b, queue[:, j]>/T     (model/pretrain.py:175-182)
//
// The contraction the north star names: M = batch (<= 32 rows per tile), N = K queue columns,
// reduction over D = 128.  12.8 FLOP per byte of queue -> HBM / latency bound: ONE launch, one
// workgroup per 64 queue columns (256 workgroups at K = 16384, one per CU), no split-K workspace.
//   * q (16 KB) goes through LDS once per workgroup (coalesced 16-byte loads);
//   * the queue tile is read straight into MFMA operand registers: lane (k-half, n) of
//     v_mfma_f32_32x32x2_f32 takes queue[d][n0+n], so a half-wave reads 128 contiguous bytes of a
//     queue row -- no LDS round trip, no barrier on the streaming operand;
//   * the four waves split D (32 channels each), meet through 32 KB of LDS and store the
//     [32][64] tile scaled by 1/T; workgroup 0 also emits the positive column.
// ---------------------------------------------------------------------------
template <int D>
__global__ void __launch_bounds__(256)
nce_logits_kernel(const float* __restrict__ q, const float* __restrict__ kpos,
                  const float* __restrict__ queue, float* __restrict__ logits, int B, int K,
                  float inv_T) {
  constexpr int NT = 64, DW = D / 4, STEPS = DW / 2, LDQ = D + 1;
  __shared__ float qs[32 * LDQ];
  __shared__ float part[4][32][NT + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int n0 = blockIdx.x * NT, m0 = blockIdx.y * 32;

  // this wave's slice of the queue tile goes straight to MFMA operand registers; every global
  // load of the workgroup (queue slice, positive keys, q) is issued before anything waits
  const int d0 = wave * DW;
  float bv[2][STEPS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + t * 32 + l31;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      bv[t][s] = col < K ? queue[(long)(d0 + 2 * s + half) * K + col] : 0.f;
  }
  float kv[8][D / 64];
  if (blockIdx.x == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int j = 0; j < D / 64; ++j)
        kv[r][j] = m0 + wave * 8 + r < B ? kpos[(long)(m0 + wave * 8 + r) * D + lane + 64 * j] : 0.f;
  }
  // q tile -> LDS (rows past B are zero)
  {
    float4 qv[32 * D / 4 / 256];
#pragma unroll
    for (int i = 0; i < 32 * D / 4 / 256; ++i) {
      const int e = tid + i * 256;
      const int r = e / (D / 4), c4 = e - r * (D / 4);
      qv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < B) qv[i] = reinterpret_cast<const float4*>(q + (long)(m0 + r) * D)[c4];
    }
#pragma unroll
    for (int i = 0; i < 32 * D / 4 / 256; ++i) {
      const int e = tid + i * 256;
      const int r = e / (D / 4), c4 = e - r * (D / 4);
      float* d = &qs[r * LDQ + c4 * 4];
      d[0] = qv[i].x; d[1] = qv[i].y; d[2] = qv[i].z; d[3] = qv[i].w;
    }
  }
  __syncthreads();
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const float av = qs[l31 * LDQ + d0 + 2 * s + half];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[0][s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[1][s], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = (i & 3) + 8 * (i >> 2) + 4 * half;
    part[wave][r][l31] = acc0[i];
    part[wave][r][32 + l31] = acc1[i];
  }
  __syncthreads();
  for (int e = tid; e < 32 * NT; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (m0 + r < B && n0 + c < K) {
      const float v = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
      logits[(long)(m0 + r) * (1 + K) + 1 + n0 + c] = v * inv_T;
    }
  }
  if (blockIdx.x == 0) {
    // positive column: wave w owns rows 8w..8w+7, keys already in registers
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = wave * 8 + r;
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < D / 64; ++j) sacc += qs[row * LDQ + lane + 64 * j] * kv[r][j];
      sacc = wave_sum(sacc);
      if (lane == 0 && m0 + row < B) logits[(long)(m0 + row) * (1 + K)] = sacc * inv_T;
    }
  }
}

// dq[b][:] += dlogits[b][0] * inv_T * k[b][:]
__global__ void lpos_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ k,
                                float* dq, int rows, int D, long ldl, float inv_T) {
  const long total = (long)rows * D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const long b = e / D;
    dq[e] = ep_lpos(dq[e], dlogits[b * ldl], inv_T, k[e]);
  }
}


// ---------------------------------------------------------------------------
// queue[:, ptr:ptr+BW] = keys^T   (queue is [D][K], keys is [BW][D])
// ---------------------------------------------------------------------------
__global__ void queue_enqueue_kernel(float* queue, const float* __restrict__ keys, int D, int K,
                                     int BW, const int64_t* __restrict__ ptr) {
  const int p = (int)(*ptr);
  // A pointer that is not a multiple of this batch (checkpoint resumed with another global batch)
  // would run past the row / the buffer; the reference raises on the slice assignment
  // (pretrain.py:93).  The host validates the pointer once per load; the kernel never writes
  // outside [0, K) whatever it holds.
  if (p < 0 || p + BW > K) return;
  const int total = D * BW;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int d = e / BW, j = e - d * BW;   // j fastest: contiguous writes along a queue row
    queue[(long)d * K + p + j] = keys[(long)j * D + d];
  }
}

__global__ void queue_fill_i64_kernel(int64_t* q, const int64_t* __restrict__ vals, int64_t cval,
                                      int K, int BW, const int64_t* __restrict__ ptr) {
  const int p = (int)(*ptr);
  if (p < 0 || p + BW > K) return;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < BW; j += gridDim.x * blockDim.x)
    q[p + j] = vals ? vals[j] : cval;
}

__global__ void queue_advance_kernel(int64_t* ptr, int BW, int K) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *ptr = (*ptr + BW) % K;
}

// ---------------------------------------------------------------------------
// mask[b][0] = 1; mask[b][1+j] = (src[b] == names[j]) | (j in top-k of sim[b] with
// the same-source entries excluded).  One block per row, row cached in LDS.
// Ties resolve to the lowest column (torch.topk leaves tie order unspecified).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
positive_mask_kernel(const float* __restrict__ sim, const int64_t* __restrict__ src,
                     const int64_t* __restrict__ names, uint8_t* __restrict__ mask, int K,
                     int topk) {
  extern __shared__ float row[];   // K floats
  __shared__ float wbest[4];
  __shared__ int widx[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t s = src[b];
  uint8_t* mrow = mask + (long)b * (1 + K);
  if (tid == 0) mrow[0] = 1;
  for (int j = tid; j < K; j += 256) {
    const bool same = names[j] == s;
    mrow[1 + j] = same ? 1 : 0;
    if (topk > 0) row[j] = same ? -INFINITY : sim[(long)b * K + j];
  }
  for (int t = 0; t < topk; ++t) {
    __syncthreads();
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < K; j += 256) {
      const float v = row[j];
      // NaN-free inputs assumed (unit vectors); prefer lower index on ties
      if (v > best || (v == best && j < bi)) { best = v; bi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((tid & 63) == 0) { wbest[tid >> 6] = best; widx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (wbest[w] > best || (wbest[w] == best && widx[w] < bi)) { best = wbest[w]; bi = widx[w]; }
      if (bi < K) {
        mrow[1 + bi] = 1;
        // NaN marks "already taken": never compares greater, unlike -inf which
        // must stay selectable when fewer than topk finite candidates remain
        row[bi] = __builtin_nanf("");
      }
    }
  }
}

// ---------------------------------------------------------------------------
// CoCLR positive mining (model/pretrain.py:397-413) as ONE launch with no (B, K) similarity
// tensor: sim = kf @ queue_second on the MFMA pipe, tile by tile, with a running top-k.
//   * a workgroup owns 32 rows x 64 queue columns: the queue tile goes straight into MFMA operand
//     registers, the four waves split the 128 feature channels and meet through LDS (the layout of
//     nce_logits_kernel);
//   * it writes the same-source ("sibling") bits of its columns into the mask and, per row, its
//     `topk` best non-sibling columns (value desc, column asc) into a candidate table;
//   * the LAST workgroup of a row tile to finish (one device-scope counter per row tile, reset by its
//     last user) merges the K/64 x topk candidates of each of its rows and sets the winners' bits.
// -inf entries need no candidates: the only -inf columns are siblings, whose bits are set anyway
// (torch.topk falls back to them, lowest index first, when fewer than topk finite values remain).
// ---------------------------------------------------------------------------
struct MineBest { float v; int i; };

__device__ __forceinline__ bool mine_better(float v, int i, float bv, int bi) {
  return v > bv || (v == bv && i < bi);
}

template <int D>
__global__ void __launch_bounds__(256)
mine_positives_kernel(const float* __restrict__ kf, const float* __restrict__ queue2,
                      const int64_t* __restrict__ src, const int64_t* __restrict__ names,
                      uint8_t* __restrict__ mask, float* cand_val, int* cand_idx, int* counters,
                      float* __restrict__ sim_out, int B, int K, int topk) {
  constexpr int NT = 64, DW = D / 4, STEPS = DW / 2, LDQ = D + 1;
  __shared__ float qs[32 * LDQ];
  __shared__ float part[4][32][NT + 1];
  __shared__ int last_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int ntiles = gridDim.x;
  const int n0 = blockIdx.x * NT, m0 = blockIdx.y * 32;

  const int d0 = wave * DW;
  float bv[2][STEPS];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int col = n0 + t * 32 + l31;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
      bv[t][s] = col < K ? queue2[(long)(d0 + 2 * s + half) * K + col] : 0.f;
  }
  {
    float4 qv[32 * D / 4 / 256];
#pragma unroll
    for (int i = 0; i < 32 * D / 4 / 256; ++i) {
      const int e = tid + i * 256;
      const int r = e / (D / 4), c4 = e - r * (D / 4);
      qv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m0 + r < B) qv[i] = reinterpret_cast<const float4*>(kf + (long)(m0 + r) * D)[c4];
    }
#pragma unroll
    for (int i = 0; i < 32 * D / 4 / 256; ++i) {
      const int e = tid + i * 256;
      const int r = e / (D / 4), c4 = e - r * (D / 4);
      float* d = &qs[r * LDQ + c4 * 4];
      d[0] = qv[i].x; d[1] = qv[i].y; d[2] = qv[i].z; d[3] = qv[i].w;
    }
  }
  __syncthreads();
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
#pragma unroll
  for (int s = 0; s < STEPS; ++s) {
    const float av = qs[l31 * LDQ + d0 + 2 * s + half];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[0][s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[1][s], acc1, 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = (i & 3) + 8 * (i >> 2) + 4 * half;
    part[wave][r][l31] = acc0[i];
    part[wave][r][32 + l31] = acc1[i];
  }
  __syncthreads();
  // thread = (row r, octet o): 8 columns of the row; siblings -> mask bit and -inf
  const int r = tid >> 3, o = tid & 7;
  const int row = m0 + r;
  const bool rok = row < B;
  const int64_t s_row = rok ? src[row] : 0;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = o * 8 + j, col = n0 + c;
    float x = (part[0][r][c] + part[1][r][c]) + (part[2][r][c] + part[3][r][c]);
    bool same = false;
    if (rok && col < K) {
      same = names[col] == s_row;
      mask[(long)row * (1 + K) + 1 + col] = same ? 1 : 0;
      if (sim_out) sim_out[(long)row * K + col] = x;
    }
    v[j] = (rok && col < K && !same) ? x : -INFINITY;
  }
  if (blockIdx.x == 0 && o == 0 && rok) mask[(long)row * (1 + K)] = 1;
  // the row's topk best of this tile: repeated arg-max over the 8 lanes of the row
  for (int t = 0; t < topk; ++t) {
    MineBest b = {-INFINITY, 0x7fffffff};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (mine_better(v[j], n0 + o * 8 + j, b.v, b.i)) { b.v = v[j]; b.i = n0 + o * 8 + j; }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      const float ov = __shfl_xor(b.v, off);
      const int oi = __shfl_xor(b.i, off);
      if (mine_better(ov, oi, b.v, b.i)) { b.v = ov; b.i = oi; }
    }
    if (b.v == -INFINITY) b.i = -1;                     // nothing (finite) left in this tile
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (n0 + o * 8 + j == b.i) v[j] = -INFINITY;       // taken
    if (o == 0 && rok) {
      const long slot = ((long)row * ntiles + blockIdx.x) * topk + t;
      cand_val[slot] = b.v;
      cand_idx[slot] = b.i;
    }
  }
  // ---- who is last for this row tile? -------------------------------------------------------
  __threadfence();                                        // candidates + mask bytes visible device-wide
  __syncthreads();
  if (tid == 0) {
    const int prev = atomicAdd(&counters[blockIdx.y], 1);
    last_flag = prev == ntiles - 1;
    if (last_flag) counters[blockIdx.y] = 0;              // ready for the next launch
  }
  __syncthreads();
  if (!last_flag || topk == 0) return;
  __threadfence();
  // ---- merge: the row's 8 lanes scan ntiles*topk candidates, topk rounds --------------------
  if (!rok) return;
  const int ncand = ntiles * topk;
  const float* cv = cand_val + (long)row * ncand;
  const int* ci = cand_idx + (long)row * ncand;
  float last_v = INFINITY;
  int last_i = -1;
  for (int t = 0; t < topk; ++t) {
    // best candidate strictly after the previous pick in (value desc, column asc) order
    MineBest b = {-INFINITY, 0x7fffffff};
    for (int e = o; e < ncand; e += 8) {
      const float x = __builtin_nontemporal_load(&cv[e]);
      const int i = __builtin_nontemporal_load(&ci[e]);
      if (i < 0) continue;
      const bool after = x < last_v || (x == last_v && i > last_i);
      if (after && mine_better(x, i, b.v, b.i)) { b.v = x; b.i = i; }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      const float ov = __shfl_xor(b.v, off);
      const int oi = __shfl_xor(b.i, off);
      if (mine_better(ov, oi, b.v, b.i)) { b.v = ov; b.i = oi; }
    }
    if (b.i == 0x7fffffff) break;                         // fewer than topk non-sibling columns
    if (o == 0) mask[(long)row * (1 + K) + 1 + b.i] = 1;
    last_v = b.v; last_i = b.i;
  }
}

// out[i][:] = in[idx[i]][:]   (rows of `row_elems` floats)
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx, float* out,
                   long row_elems, long in_row_stride) {
  const long r = blockIdx.y;
  const float* src = in + idx[r] * in_row_stride;
  float* dst = out + r * row_elems;
  if (((row_elems | in_row_stride) & 3) == 0) {
    const long n4 = row_elems >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_elems; i += (long)gridDim.x * 256)
      dst[i] = src[i];
  }
}

// out[i][:] = *rows[i]: every output row has its own source ADDRESS -- local memory or a peer GPU's
// buffer mapped through hipIpc (xGMI reads): the shuffle-BN exchange as a row pull, each rank
// fetching exactly the B clips it will encode (model/pretrain.py:98-124 gathers all B*world).
__global__ void __launch_bounds__(256)
pull_rows_kernel(const int64_t* __restrict__ rows, float* out, long row_elems) {
  const long r = blockIdx.y;
  const float* src = reinterpret_cast<const float*>(rows[r]);
  float* dst = out + r * row_elems;
  if ((row_elems & 3) == 0 && (((uintptr_t)src) & 15) == 0) {
    const long n4 = row_elems >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
      reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_elems; i += (long)gridDim.x * 256)
      dst[i] = src[i];
  }
}

__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    y[e] = fmaxf(x[e], 0.f);
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dx, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    dx[e] = y[e] > 0.f ? dy[e] : 0.f;
}

// out[c] = sum_r x[r][c]
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                              int cols) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = 0; r < rows; ++r) s += x[(long)r * cols + c];
  out[c] = s;
}

inline int grid1d(long n, int cap = 2048) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int coclr_gemm_workspace(int M, int N, int K, int splits, int64_t* elems) {
  *elems = splits > 1 ? (int64_t)splits * M * N : 0;
  return 0;
}

namespace {
int launch_gemm(GemmArgs& g, int splits, hipStream_t stream) {
  int kslice = cdiv(g.K, splits);
  kslice = cdiv(kslice, 32) * 32;
  g.kslice = kslice;
  splits = cdiv(g.K, kslice);
  g.splits = splits;
  if (splits > 1) g.to_part = 1;
  dim3 grid(cdiv(g.N, 128), cdiv(g.M, 32), splits);
  const bool TA = (g.sam == 1 && g.sak != 1), TB = (g.sbk == 1 && g.sbn != 1);
  if (TA && TB) hipLaunchKernelGGL((gemm32x128_kernel<true, true>), grid, dim3(256), 0, stream, g);
  else if (TA) hipLaunchKernelGGL((gemm32x128_kernel<true, false>), grid, dim3(256), 0, stream, g);
  else if (TB) hipLaunchKernelGGL((gemm32x128_kernel<false, true>), grid, dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm32x128_kernel<false, false>), grid, dim3(256), 0, stream, g);
  COCLR_LAUNCH_CHECK();
  if (!g.to_part) return 0;
  if (g.ep_mode == 2 || g.ep_mode == 3) {
    const dim3 rgrid(cdiv((long)g.M * 64, 256));
    if (g.N <= 128) hipLaunchKernelGGL((gemm_splitk_reduce_rows_kernel<2>), rgrid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((gemm_splitk_reduce_rows_kernel<8>), rgrid, dim3(256), 0, stream, g);
  } else if (g.ep_mode == 4) {
    hipLaunchKernelGGL(gemm_splitk_reduce_expand_kernel, dim3(grid1d((long)g.M * g.N * g.ep_S, 8192)),
                       dim3(256), 0, stream, g);
  } else {
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(grid1d((long)g.M * g.N)), dim3(256), 0, stream, g);
  }
  COCLR_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int coclr_gemm(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk,
                          int64_t sbn, float* c, int64_t ldc, const float* bias, int M, int N,
                          int K, float alpha, int relu, int accumulate, int splits,
                          float* workspace, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0 || K <= 0 || splits < 1) return COCLR_EINVAL;
  if (splits > 1 && !workspace) return COCLR_EINVAL;
  GemmArgs g = {};
  g.a = a; g.b = b; g.c = c; g.bias = bias;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.relu = relu; g.accumulate = accumulate; g.part = workspace;
  return launch_gemm(g, splits, stream);
}

// The same product with the row-level operation that follows it in the projection head applied by the fold
// kernel (model/pretrain.py:49-54,153-154,175-182 and their backward): two launches per product, one for the
// plain product with row sums.
extern "C" int coclr_gemm_fused(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk,
                                int64_t sbn, float* c, int64_t ldc, const float* bias, int M, int N,
                                int K, float alpha, int relu, int splits, float* workspace,
                                const coclr_gemm_epilogue* ep, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (M <= 0 || N <= 0 || K <= 0 || splits < 1 || !ep) return COCLR_EINVAL;
  if (ep->mode < 0 || ep->mode > 4) return COCLR_EINVAL;
  const bool direct = ep->mode == 0 && splits == 1;            // plain product (+ row sums), one launch
  if (!direct && !workspace) return COCLR_EINVAL;
  if ((ep->mode == 2 || ep->mode == 3) && N > 512) return COCLR_EINVAL;      // a row lives in one wave
  if (ep->mode == 1 && !ep->a) return COCLR_EINVAL;
  if (ep->mode == 3 && (!ep->a || !ep->b || !ep->y || !ep->inv_norm)) return COCLR_EINVAL;
  if (ep->mode == 4 && ep->S <= 0) return COCLR_EINVAL;
  if (ep->rowsum && splits != 1) return COCLR_EINVAL;
  GemmArgs g = {};
  g.a = a; g.b = b; g.c = c; g.bias = bias;
  g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc;
  g.M = M; g.N = N; g.K = K;
  g.alpha = alpha; g.relu = relu; g.accumulate = 0; g.part = workspace;
  g.to_part = direct ? 0 : 1;
  g.ep_mode = ep->mode; g.ep_S = ep->S;
  g.ep_a = ep->a; g.ep_lda = ep->lda; g.ep_b = ep->b; g.ep_y = ep->y; g.ep_inv = ep->inv_norm;
  g.ep_out2 = ep->out2; g.ep_f = ep->f; g.rowsum = ep->rowsum;
  return launch_gemm(g, splits, stream);
}

extern "C" int coclr_l2norm_fwd(const float* x, float* y, float* inv_norm, int rows, int D,
                                float eps, void* stream) {
  if (rows <= 0 || D <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(cdiv((long)rows * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, inv_norm, rows, D, eps);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx,
                                int rows, int D, void* stream) {
  if (rows <= 0 || D <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(cdiv((long)rows * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, y, inv_norm, dx, rows, D);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// logits[B][1+K] = [ <q,k> | q . queue ] / T      (model/pretrain.py:175-182)
extern "C" int coclr_nce_logits_fwd(const float* q, const float* k, const float* queue,
                                    float* logits, int B, int D, int K, float T, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || D <= 0 || K <= 0 || T == 0.f) return COCLR_EINVAL;
  const float inv_T = 1.f / T;
  if (D == 128 && ((uintptr_t)q & 15) == 0) {
    // the head's own shape (dim = 128, model/pretrain.py:32): one fused launch
    hipLaunchKernelGGL((nce_logits_kernel<128>), dim3(cdiv(K, 64), cdiv(B, 32)), dim3(256), 0, stream,
                       q, k, queue, logits, B, K, inv_T);
    COCLR_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(lpos_fwd_kernel, dim3(cdiv((long)B * 64, 256)), dim3(256), 0, stream, q, k,
                     logits, B, D, (long)(1 + K), inv_T);
  COCLR_LAUNCH_CHECK();
  return coclr_gemm(q, D, 1, queue, K, 1, logits + 1, 1 + K, nullptr, B, K, D, inv_T, 0, 0, 1,
                    nullptr, stream_);
}

// dq[B][D] = ( dlogits[:,1:] . queue^T + dlogits[:,0] * k ) / T
extern "C" int coclr_nce_logits_bwd(const float* dlogits, const float* k, const float* queue,
                                    float* dq, float* workspace, int B, int D, int K, float T,
                                    int splits, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || D <= 0 || K <= 0 || T == 0.f) return COCLR_EINVAL;
  const float inv_T = 1.f / T;
  // A(m,k') = dlogits[m][1+k'] (k' contiguous), B(k',n) = queue[n][k'] (k' contiguous)
  int rc = coclr_gemm(dlogits + 1, 1 + K, 1, queue, 1, K, dq, D, nullptr, B, D, K, inv_T, 0, 0,
                      splits, workspace, stream_);
  if (rc) return rc;
  hipLaunchKernelGGL(lpos_bwd_kernel, dim3(grid1d((long)B * D)), dim3(256), 0, stream, dlogits, k,
                     dq, B, D, (long)(1 + K), inv_T);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_queue_enqueue(float* queue, const float* keys, int D, int K, int BW,
                                   const int64_t* ptr, void* stream) {
  if (D <= 0 || K <= 0 || BW <= 0 || BW > K || K % BW != 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(queue_enqueue_kernel, dim3(grid1d((long)D * BW)), dim3(256), 0,
                     (hipStream_t)stream, queue, keys, D, K, BW, ptr);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_queue_fill_i64(int64_t* queue, const int64_t* vals, int64_t const_val, int K,
                                    int BW, const int64_t* ptr, void* stream) {
  if (K <= 0 || BW <= 0 || BW > K) return COCLR_EINVAL;
  hipLaunchKernelGGL(queue_fill_i64_kernel, dim3(grid1d(BW)), dim3(256), 0, (hipStream_t)stream,
                     queue, vals, const_val, K, BW, ptr);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_queue_advance(int64_t* ptr, int BW, int K, void* stream) {
  if (K <= 0 || BW <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(queue_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ptr, BW, K);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_positive_mask(const float* sim, const int64_t* src, const int64_t* names,
                                   uint8_t* mask, int B, int K, int topk, void* stream) {
  if (B <= 0 || K <= 0 || topk < 0 || topk > K) return COCLR_EINVAL;
  if (topk > 0 && !sim) return COCLR_EINVAL;
  const size_t lds = topk > 0 ? (size_t)K * sizeof(float) : 0;
  if (lds > 150 * 1024) return COCLR_EINVAL;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(positive_mask_kernel), 150 * 1024, attr_done));
  hipLaunchKernelGGL(positive_mask_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, sim, src,
                     names, mask, K, topk);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_mine_positives(const float* kf, const float* queue_second, const int64_t* src,
                                    const int64_t* names, uint8_t* mask, float* cand_val,
                                    int32_t* cand_idx, int32_t* counters, float* sim_out, int B,
                                    int D, int K, int topk, void* stream) {
  if (!kf || !queue_second || !src || !names || !mask || !counters) return COCLR_EINVAL;
  if (B <= 0 || K <= 0 || D != 128 || topk < 0 || topk > 16 || topk > K) return COCLR_EINVAL;
  if (topk > 0 && (!cand_val || !cand_idx)) return COCLR_EINVAL;
  // the row-tile counters must be zero on entry; a launch that was aborted, or a workspace another
  // stream is still using, would otherwise leave them non-zero and every later merge would fire early or
  // never: zero them in stream order (B/32 ints).  The workspace is single-stream.
  COCLR_RETURN_IF(hipMemsetAsync(counters, 0, sizeof(int32_t) * (size_t)cdiv(B, 32), (hipStream_t)stream));
  hipLaunchKernelGGL(mine_positives_kernel<128>, dim3(cdiv(K, 64), cdiv(B, 32)), dim3(256), 0,
                     (hipStream_t)stream, kf, queue_second, src, names, mask, cand_val, cand_idx,
                     counters, sim_out, B, K, topk);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_gather_rows(const float* in, const int64_t* idx, float* out, int rows,
                                 int64_t row_elems, int64_t in_row_stride, void* stream) {
  if (rows <= 0 || row_elems <= 0 || in_row_stride < row_elems) return COCLR_EINVAL;
  int gx = (int)((row_elems / 4 + 255) / 256);
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, in, idx,
                     out, (long)row_elems, (long)in_row_stride);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_pull_rows(const int64_t* row_ptrs, float* out, int rows, int64_t row_elems,
                               void* stream) {
  if (rows <= 0 || row_elems <= 0 || !row_ptrs || !out) return COCLR_EINVAL;
  int gx = (int)((row_elems / 4 + 255) / 256);
  if (gx < 1) gx = 1;
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(pull_rows_kernel, dim3(gx, rows), dim3(256), 0, (hipStream_t)stream, row_ptrs,
                     out, (long)row_elems);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// ---- S3D-G self gating (backbone/s3dg.py:68-78): out = x * sigmoid(fc(mean(x))) ----------
namespace {

// w = sigmoid(s); ds = dw * w * (1 - w) for the backward
__global__ void sigmoid_fwd_kernel(const float* __restrict__ s, float* __restrict__ w, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    w[e] = 1.f / (1.f + expf(-s[e]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ w,
                                   float* __restrict__ ds, long n) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
    ds[e] = dw[e] * w[e] * (1.f - w[e]);
}

// out[n][c][:] (+)= a[n][c][:] * g[n*C+c] + b[n*C+c]   (one wave per plane, 16 B lanes)
__global__ void __launch_bounds__(256)
plane_scale_kernel(const float* __restrict__ a, const float* __restrict__ g,
                   const float* __restrict__ b, float* out, int planes, int C, int S,
                   long a_nstride, long out_nstride, int accumulate) {
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pl >= planes) return;
  const int n = pl / C, c = pl - n * C;
  const float gv = g[pl], bv = b ? b[pl] : 0.f;
  const float* ap = a + (long)n * a_nstride + (long)c * S;
  float* op = out + (long)n * out_nstride + (long)c * S;
  for (int i = lane; i < S; i += 64) {
    const float v = fmaf(ap[i], gv, bv);
    op[i] = accumulate ? op[i] + v : v;
  }
}

// out[n*C+c] = sum_s a[n][c][s] * b[n][c][s]   (one wave per plane)
__global__ void __launch_bounds__(256)
plane_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                 int planes, int C, int S, long a_nstride, long b_nstride) {
  const int pl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pl >= planes) return;
  const int n = pl / C, c = pl - n * C;
  const float* ap = a + (long)n * a_nstride + (long)c * S;
  const float* bp = b + (long)n * b_nstride + (long)c * S;
  float acc = 0.f;
  for (int i = lane; i < S; i += 64) acc = fmaf(ap[i], bp[i], acc);
  acc = wave_sum(acc);
  if (lane == 0) out[pl] = acc;
}

}  // namespace

extern "C" int coclr_sigmoid_fwd(const float* s, float* w, int64_t n, void* stream) {
  if (n <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, s, w,
                     (long)n);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_sigmoid_bwd(const float* dw, const float* w, float* ds, int64_t n,
                                 void* stream) {
  if (n <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dw, w,
                     ds, (long)n);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_plane_scale(const float* a, const float* gain, const float* bias, float* out,
                                 int N, int C, int64_t S, int64_t a_nstride, int64_t out_nstride,
                                 int accumulate, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return COCLR_EINVAL;
  const int planes = N * C;
  hipLaunchKernelGGL(plane_scale_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream,
                     a, gain, bias, out, planes, C, (int)S, (long)a_nstride, (long)out_nstride,
                     accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_plane_dot(const float* a, const float* b, float* out, int N, int C, int64_t S,
                               int64_t a_nstride, int64_t b_nstride, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return COCLR_EINVAL;
  const int planes = N * C;
  hipLaunchKernelGGL(plane_dot_kernel, dim3(cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, a,
                     b, out, planes, C, (int)S, (long)a_nstride, (long)b_nstride);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
  if (n <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, x, y,
                     (long)n);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
  if (n <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx,
                     (long)n);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_colsum(const float* x, float* out, int rows, int cols, void* stream) {
  if (rows <= 0 || cols <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                     rows, cols);
  COCLR_LAUNCH_CHECK();
  return 0;
}
