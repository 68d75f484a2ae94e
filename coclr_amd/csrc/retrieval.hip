// Evaluation consumers of the backbone for gfx950:
//   * LinearClassifier head (model/classifier.py:47-61): BatchNorm1d statistics of the pooled
//     (N, C) features (final_bn) -- the rest of the head reuses the GEMM / l2norm / BN kernels
//   * nearest-neighbour retrieval (eval/main_classifier.py:686-706): centre and normalise the
//     feature matrices, sim = test . train^T, then for k in (1,5,10,20,50) "is any of the k most
//     similar training clips of the test clip's class": one kernel selects the top-kmax of every
//     similarity row IN ORDER and compares labels on the way, instead of five torch.topk calls
//     over the full (n_test, n_train) matrix
#include "common.h"
#include "../../include/coclr_hip.h"
#include <math.h>

namespace {

// partial[r][c] = sum over rows r, r+R, ... of x[row][c]  (and of x^2 when `sq`)
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                      float* __restrict__ part_sq, int rows, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int R = gridDim.y, r0 = blockIdx.y;
  float s = 0.f, q = 0.f;
  for (int r = r0; r < rows; r += R) {
    const float v = x[(long)r * cols + c];
    s += v;
    q = fmaf(v, v, q);
  }
  part[(long)r0 * cols + c] = s;
  if (part_sq) part_sq[(long)r0 * cols + c] = q;
}

// out[r][c] = x[r][c] - mean[c], mean from the R partial sums (fp64 fold)
__global__ void __launch_bounds__(256)
center_rows_kernel(const float* __restrict__ x, const float* __restrict__ part, float* __restrict__ out,
                   int rows, int cols, int R) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  double s = 0.0;
  for (int r = 0; r < R; ++r) s += part[(long)r * cols + c];
  const float mean = (float)(s / rows);
  for (int r = blockIdx.y; r < rows; r += gridDim.y)
    out[(long)r * cols + c] = x[(long)r * cols + c] - mean;
}

// stats[0][c] = sum_r x[r][c], stats[1][c] = sum_r x[r][c]^2   (one "tile" per channel: the
// layout coclr_bn_finalize takes with ntiles = 1)
__global__ void __launch_bounds__(256)
bn1d_fold_kernel(const float* __restrict__ part, const float* __restrict__ part_sq,
                 float* __restrict__ stats, int cols, int R) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  double s = 0.0, q = 0.0;
  for (int r = 0; r < R; ++r) { s += part[(long)r * cols + c]; q += part_sq[(long)r * cols + c]; }
  stats[c] = (float)s;
  stats[cols + c] = (float)q;
}

// Ordered top-kmax of one similarity row; hits[b][i] = any(train_label[top ks[i]] == test_label[b]).
// Selection order is (value descending, column ascending); the t-th pick is the largest element
// that comes strictly after the (t-1)-th in that order, so nothing is marked or modified.
__global__ void __launch_bounds__(256)
retrieval_hits_kernel(const float* __restrict__ sim, const int64_t* __restrict__ train_label,
                      const int64_t* __restrict__ test_label, const int32_t* __restrict__ ks, int nk,
                      int kmax, float* __restrict__ hits, int32_t* __restrict__ topidx, int N,
                      int use_lds) {
  extern __shared__ float rowbuf[];
  __shared__ float wbest[4];
  __shared__ int widx[4];
  __shared__ float s_pv;
  __shared__ int s_pi;
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = sim + (long)b * N;
  if (use_lds) {
    for (int j = tid; j < N; j += 256) rowbuf[j] = row[j];
  }
  const int64_t want = test_label[b];
  if (tid == 0) { s_pv = INFINITY; s_pi = -1; }
  bool found = false;      // thread 0 only
  int next_k = 0;
  __syncthreads();
  for (int t = 0; t < kmax; ++t) {
    const float pv = s_pv;
    const int pi = s_pi;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < N; j += 256) {
      const float v = use_lds ? rowbuf[j] : row[j];
      const bool eligible = (v < pv) || (v == pv && j > pi);
      if (eligible && (v > best || (v == best && j < bi))) { best = v; bi = j; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int oi = __shfl_xor(bi, off);
      if (oi != 0x7fffffff && (bi == 0x7fffffff || ob > best || (ob == best && oi < bi))) {
        best = ob; bi = oi;
      }
    }
    if ((tid & 63) == 0) { wbest[tid >> 6] = best; widx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (widx[w] != 0x7fffffff &&
            (bi == 0x7fffffff || wbest[w] > best || (wbest[w] == best && widx[w] < bi))) {
          best = wbest[w]; bi = widx[w];
        }
      if (bi != 0x7fffffff) {
        if (train_label[bi] == want) found = true;
        if (topidx) topidx[(long)b * kmax + t] = bi;
        s_pv = best; s_pi = bi;
      } else if (topidx) {
        topidx[(long)b * kmax + t] = -1;
      }
      while (next_k < nk && ks[next_k] == t + 1) { hits[(long)b * nk + next_k] = found ? 1.f : 0.f; ++next_k; }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int coclr_colstats_workspace(int rows, int cols, int64_t* elems) {
  if (rows <= 0 || cols <= 0) return COCLR_EINVAL;
  int R = rows < 64 ? rows : 64;
  *elems = (int64_t)2 * R * cols;
  return 0;
}

static int row_splits(int rows) { return rows < 64 ? rows : 64; }

// BatchNorm1d batch statistics of x[rows][cols] as [2][cols] (= coclr_bn_finalize, ntiles 1)
extern "C" int coclr_bn1d_stats(const float* x, float* stats, float* workspace, int rows, int cols,
                                void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0 || cols <= 0 || !x || !stats || !workspace) return COCLR_EINVAL;
  const int R = row_splits(rows);
  float* part = workspace;
  float* part_sq = workspace + (long)R * cols;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 256), R), dim3(256), 0, stream, x, part,
                     part_sq, rows, cols);
  COCLR_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn1d_fold_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, stream, part, part_sq,
                     stats, cols, R);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// out = x - x.mean(0)   (eval/main_classifier.py:690-691)
extern "C" int coclr_center_rows(const float* x, float* out, float* workspace, int rows, int cols,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (rows <= 0 || cols <= 0 || !x || !out || !workspace) return COCLR_EINVAL;
  const int R = row_splits(rows);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(cdiv(cols, 256), R), dim3(256), 0, stream, x,
                     workspace, (float*)nullptr, rows, cols);
  COCLR_LAUNCH_CHECK();
  int gy = rows < 256 ? rows : 256;
  hipLaunchKernelGGL(center_rows_kernel, dim3(cdiv(cols, 256), gy), dim3(256), 0, stream, x,
                     workspace, out, rows, cols, R);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_retrieval_hits(const float* sim, const int64_t* train_label,
                                    const int64_t* test_label, const int32_t* ks, int nk,
                                    float* hits, int32_t* topidx, int B, int N, int kmax,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || N <= 0 || nk <= 0 || kmax <= 0 || kmax > N || !sim || !ks || !hits) return COCLR_EINVAL;
  auto kern = retrieval_hits_kernel;
  const size_t lds_row = (size_t)N * sizeof(float);
  const int use_lds = lds_row <= 150 * 1024;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 150 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), use_lds ? lds_row : 0, stream, sim, train_label,
                     test_label, ks, nk, kmax, hits, topidx, N, use_lds);
  COCLR_LAUNCH_CHECK();
  return 0;
}
