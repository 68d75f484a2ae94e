// Loss + accuracy epilogue over the (B, 1+K) logits for gfx950 -- what the launch scripts compute
// with 8-12 ATen ops and 3-5 .item() syncs per step:
//   InfoNCE  main_nce.py:314-316   CrossEntropyLoss(output, target) + calc_topk_accuracy(1,5)
//   UberNCE  main_nce.py:318-324   -(log_softmax*mask).sum(1)/mask.sum(1), calc_mask_accuracy(1,5)
//   CoCLR    main_coclr.py:343-346,388-401  multi_nce_loss = -log((softmax*mask).sum(1)) with the
//            self-similarity column dropped for rows that have other positives, calc_mask_accuracy
//            and calc_topk_accuracy against column 0   (utils/utils.py:52-85)
// One workgroup per row reads the row twice (max, then exp-sums / counts; the second pass hits
// L2), a one-workgroup tail folds the B rows into five device scalars.  Nothing comes back to
// the host.  Top-k hits are decided by RANK: the best positive is among the k largest logits of
// its row iff fewer than k logits are strictly greater (ties go to the positive, as a stable
// descending sort would place the lower column first for column 0).
#include "common.h"
#include "../../include/coclr_hip.h"
#include <math.h>

namespace {

__device__ __forceinline__ float block256_max(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// row statistics: float[8] = {loss, lse, aux, hit@k1, hit@k2, self hit@k1, self hit@k2, rowmax}
//   aux = modes 0/1: log-sum-exp over the positives used by the loss; mode 2: their number.
// flags[b] bit 0: column 0 was dropped from this row's loss mask (mode 1, drop_self).
__global__ void __launch_bounds__(256)
nce_loss_rows_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                     const int64_t* __restrict__ target, float* __restrict__ rowstats,
                     uint8_t* __restrict__ flags, int N1, int mode, int drop_self, int k1, int k2) {
  __shared__ float redf[4];
  __shared__ double redd[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* row = logits + (long)b * N1;
  const uint8_t* mrow = mask ? mask + (long)b * N1 : nullptr;
  const int tcol = mode == 0 ? (int)target[b] : 0;

  // pass 1: row max, best positive, number of positives
  float vmax = -INFINITY, pmax = -INFINITY;
  int npos = 0;
  for (int j = tid; j < N1; j += 256) {
    const float v = row[j];
    vmax = fmaxf(vmax, v);
    const bool pos = mode == 0 ? (j == tcol) : (mrow[j] != 0);
    if (pos) { pmax = fmaxf(pmax, v); ++npos; }
  }
  vmax = block256_max(vmax, redf);
  pmax = block256_max(pmax, redf);
  const int npos_all = (int)(block256_sum_d((double)npos, redd) + 0.5);
  const float l0 = row[0];
  const bool drop0 = mode == 1 && drop_self && npos_all != 1 && mrow[0] != 0;

  // pass 2: exp sums and rank counts
  double se = 0.0, pe = 0.0, ps = 0.0;
  int gt_p = 0, gt_0 = 0;
  for (int j = tid; j < N1; j += 256) {
    const float v = row[j];
    const float e = expf(v - vmax);
    se += e;
    bool pos = mode == 0 ? (j == tcol) : (mrow[j] != 0);
    if (drop0 && j == 0) pos = false;
    if (pos) { pe += e; ps += v; }
    gt_p += v > pmax;
    gt_0 += v > l0;
  }
  se = block256_sum_d(se, redd);
  pe = block256_sum_d(pe, redd);
  ps = block256_sum_d(ps, redd);
  const int cgp = (int)(block256_sum_d((double)gt_p, redd) + 0.5);
  const int cg0 = (int)(block256_sum_d((double)gt_0, redd) + 0.5);
  if (tid == 0) {
    const float lse = vmax + logf((float)se);
    const int npos_eff = npos_all - (drop0 ? 1 : 0);
    float loss, aux;
    if (mode == 2) {            // -(sum_pos log_softmax) / n_pos
      aux = (float)npos_eff;
      loss = lse - (float)(ps / (double)npos_eff);
    } else {                    // -log(sum_pos softmax); mode 0 is the one-positive case
      const float lsp = vmax + logf((float)pe);
      aux = lsp;
      loss = lse - lsp;
    }
    float* o = rowstats + 8 * (long)b;
    o[0] = loss; o[1] = lse; o[2] = aux;
    o[3] = cgp < k1 ? 1.f : 0.f; o[4] = cgp < k2 ? 1.f : 0.f;
    o[5] = cg0 < k1 ? 1.f : 0.f; o[6] = cg0 < k2 ? 1.f : 0.f;
    o[7] = vmax;
    flags[b] = drop0 ? 1 : 0;
  }
}

// scalars[0..4] = mean over rows of {loss, hit@k1, hit@k2, self hit@k1, self hit@k2}
__global__ void __launch_bounds__(256)
nce_loss_fold_kernel(const float* __restrict__ rowstats, float* __restrict__ scalars, int B) {
  __shared__ double red[4];
  const int fields[5] = {0, 3, 4, 5, 6};
  for (int f = 0; f < 5; ++f) {
    double s = 0.0;
    for (int b = threadIdx.x; b < B; b += 256) s += rowstats[8 * (long)b + fields[f]];
    s = block256_sum_d(s, red);
    if (threadIdx.x == 0) scalars[f] = (float)(s / B);
  }
}

// dlogits[b][j] = dloss/B * (softmax_j - w_j)
__global__ void __launch_bounds__(256)
nce_loss_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                    const int64_t* __restrict__ target, const float* __restrict__ rowstats,
                    const uint8_t* __restrict__ flags, const float* __restrict__ dloss,
                    float* __restrict__ dlogits, int B, int N1, int mode) {
  const int b = blockIdx.y;
  const float* st = rowstats + 8 * (long)b;
  const float lse = st[1], aux = st[2];
  const float gs = dloss[0] / (float)B;
  const bool drop0 = flags[b] != 0;
  const int tcol = mode == 0 ? (int)target[b] : 0;
  const float* row = logits + (long)b * N1;
  const uint8_t* mrow = mask ? mask + (long)b * N1 : nullptr;
  float* drow = dlogits + (long)b * N1;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < N1; j += gridDim.x * 256) {
    const float v = row[j];
    float w = 0.f;
    if (mode == 0) {
      w = j == tcol ? 1.f : 0.f;
    } else if (mrow[j] != 0 && !(drop0 && j == 0)) {
      w = mode == 1 ? expf(v - aux) : 1.f / aux;
    }
    drow[j] = gs * (expf(v - lse) - w);
  }
}

}  // namespace

extern "C" int coclr_nce_loss_fwd(const float* logits, const uint8_t* mask, const int64_t* target,
                                  float* rowstats, uint8_t* flags, float* scalars, int B, int N1,
                                  int mode, int drop_self, int k1, int k2, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || N1 <= 0 || mode < 0 || mode > 2 || k1 < 1 || k2 < 1) return COCLR_EINVAL;
  if ((mode == 0 && !target) || (mode != 0 && !mask)) return COCLR_EINVAL;
  if (!rowstats || !flags || !scalars) return COCLR_EINVAL;
  hipLaunchKernelGGL(nce_loss_rows_kernel, dim3(B), dim3(256), 0, stream, logits, mask, target,
                     rowstats, flags, N1, mode, drop_self, k1, k2);
  COCLR_LAUNCH_CHECK();
  hipLaunchKernelGGL(nce_loss_fold_kernel, dim3(1), dim3(256), 0, stream, rowstats, scalars, B);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_nce_loss_bwd(const float* logits, const uint8_t* mask, const int64_t* target,
                                  const float* rowstats, const uint8_t* flags, const float* dloss,
                                  float* dlogits, int B, int N1, int mode, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (B <= 0 || N1 <= 0 || mode < 0 || mode > 2) return COCLR_EINVAL;
  if ((mode == 0 && !target) || (mode != 0 && !mask)) return COCLR_EINVAL;
  int gx = cdiv(N1, 256 * 4);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(nce_loss_bwd_kernel, dim3(gx, B), dim3(256), 0, stream, logits, mask, target,
                     rowstats, flags, dloss, dlogits, B, N1, mode);
  COCLR_LAUNCH_CHECK();
  return 0;
}
