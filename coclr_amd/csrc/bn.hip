// Train/eval BatchNorm3d (+ReLU, + optional residual add) around the conv
// kernels, NCDHW fp32.  Replaces ATen batch_norm / relu_ / their backward as
// invoked from backbone/s3dg.py:16-17,25-27,46-48,59-64 and
// backbone/resnet_2d3d.py:54-83 (eps 1e-5, momentum 0.1 defaults).
//
// All of these are HBM-bound streaming kernels: 16 B/lane loads, one
// (n, c) plane per blockIdx.y stripe, per-channel coefficients in SGPRs.
// The batch statistics themselves come for free from the conv epilogue
// (partial sums per workgroup); bn_finalize folds them in fp64.
#include "common.h"
#include "../../include/coclr_hip.h"

namespace {

// ---- forward -----------------------------------------------------------------

// One block per channel: fold [2][C][ntiles] partial sums, emit mean / invstd /
// fused scale+shift, update running stats (unbiased var), bump num_batches_tracked.
__global__ void __launch_bounds__(256)
bn_finalize_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, int C,
                   int ntiles, double count,
                   const float* __restrict__ gamma, const float* __restrict__ beta,
                   float* running_mean, float* running_var, int64_t* num_batches_tracked,
                   float momentum, float eps, float* mean_out, float* invstd_out,
                   float* scale_out, float* shift_out) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  const float* ps = sum + (long)c * ntiles;
  const float* pq = sumsq + (long)c * ntiles;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += 256) { s += (double)ps[i]; q += (double)pq[i]; }
  s = block256_sum_d(s, red);
  q = block256_sum_d(q, red);
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    scale_out[c] = sc;
    shift_out[c] = b - (float)mean * sc;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  }
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float eps, int C, float* mean_out, float* invstd_out,
                                      float* scale_out, float* shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  mean_out[c] = rm[c];
  invstd_out[c] = invstd;
  scale_out[c] = sc;
  shift_out[c] = beta[c] - rm[c] * sc;
}

// z = act(y*scale[c] + shift[c] (+ res)),  act = ReLU or identity.
// y is contiguous [N][C][S]; z/res may live inside wider tensors (channel
// slices of a concat buffer) -> explicit sample strides.
typedef float nt_f32x4 __attribute__((ext_vector_type(4)));

inline long bn_nt_bytes() {
  const char* e = getenv("COCLR_BN_NT_MB");
  const long mb = e ? atol(e) : 0;
  return mb < 0 ? -1 : (mb << 20);
}

// 16-byte load / store, optionally NON-TEMPORAL: the streaming passes over tensors far larger than the 256 MB
// infinity cache would otherwise evict what the MFMA-bound kernels of the other streams keep re-reading there
__device__ __forceinline__ float4 ld4(const float* p, int i, bool nt) {
  if (nt) {
    const nt_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(p) + i);
    return make_float4(t.x, t.y, t.z, t.w);
  }
  return reinterpret_cast<const float4*>(p)[i];
}
__device__ __forceinline__ void st4(float* p, int i, const float4& w, bool nt) {
  if (nt) {
    nt_f32x4 t; t.x = w.x; t.y = w.y; t.z = w.z; t.w = w.w;
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4*>(p) + i);
  } else {
    reinterpret_cast<float4*>(p)[i] = w;
  }
}

template <bool VEC, bool NT = false>
__global__ void __launch_bounds__(256)
bn_act_apply_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                    const float* __restrict__ shift, const float* __restrict__ res, float* z,
                    int N, int C, int S, long y_nstride, long z_nstride, long res_nstride,
                    int relu) {
  // blockIdx.y = (sample group, channel): a block walks the samples n = group, group + nsg, ...
  // of ONE channel, so small planes (S = 64..512 in the last stages) still give every block a
  // few thousand elements instead of one 2 KB plane
  const int c = blockIdx.y % C, sg = blockIdx.y / C, nsg = gridDim.y / C;
  const float sc = scale[c], sf = shift[c];
  for (int n = sg; n < N; n += nsg) {
    const float* yp = y + (long)n * y_nstride + (long)c * S;
    float* zp = z + (long)n * z_nstride + (long)c * S;
    const float* rp = res ? res + (long)n * res_nstride + (long)c * S : nullptr;
    if (VEC) {
      // four independent 16-byte loads per thread and trip: one load in flight per wave is ~32 KB per CU, short
      // of what 8 TB/s needs at the loaded latency
      const int S4 = S >> 2, stride = gridDim.x * 256;
      for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < S4; i0 += 4 * stride) {
        float4 v[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * stride;
          if (i < S4) {
            if (NT) {
              const nt_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f32x4*>(yp) + i);
              v[k] = make_float4(t.x, t.y, t.z, t.w);
            } else {
              v[k] = reinterpret_cast<const float4*>(yp)[i];
            }
            if (rp) r[k] = reinterpret_cast<const float4*>(rp)[i];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = i0 + k * stride;
          if (i < S4) {
            float4 w = v[k];
            w.x = fmaf(w.x, sc, sf); w.y = fmaf(w.y, sc, sf);
            w.z = fmaf(w.z, sc, sf); w.w = fmaf(w.w, sc, sf);
            if (rp) { w.x += r[k].x; w.y += r[k].y; w.z += r[k].z; w.w += r[k].w; }
            if (relu) {
              w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f);
              w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
            }
            if (NT) {
              nt_f32x4 t; t.x = w.x; t.y = w.y; t.z = w.z; t.w = w.w;
              __builtin_nontemporal_store(t, reinterpret_cast<nt_f32x4*>(zp) + i);
            } else {
              reinterpret_cast<float4*>(zp)[i] = w;
            }
          }
        }
      }
    } else {
      for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
        float v = fmaf(yp[i], sc, sf);
        if (rp) v += rp[i];
        if (relu) v = fmaxf(v, 0.f);
        zp[i] = v;
      }
    }
  }
}

// ---- backward ----------------------------------------------------------------

// sums[c] += (sum g, sum g*xhat), g = dz * mask.  mask = (z > 0) when z is given
// (residual units), else recomputed as (y*scale+shift > 0); no mask if !relu.
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_act_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                         const float* __restrict__ z, const float* __restrict__ scale,
                         const float* __restrict__ shift, const float* __restrict__ mean,
                         const float* __restrict__ invstd, double* sums, int N, int C, int S,
                         long dz_nstride, long y_nstride, long z_nstride, int relu_nt) {
  __shared__ double red[4];
  const int relu = relu_nt & 1;
  const bool nt = (relu_nt & 2) != 0;
  const int c = blockIdx.x;
  const float sc = scale[c], sf = shift[c], mu = mean[c], is = invstd[c];
  double sg = 0.0, sgx = 0.0;
  for (int n = blockIdx.y; n < N; n += gridDim.y) {
    const float* dzp = dz + (long)n * dz_nstride + (long)c * S;
    const float* yp = y + (long)n * y_nstride + (long)c * S;
    const float* zp = z ? z + (long)n * z_nstride + (long)c * S : nullptr;
    float ag = 0.f, agx = 0.f;
    if (VEC) {
      const int S4 = S >> 2;
      // two (dz, y) pairs in flight per thread and trip; the accumulation order over i is unchanged
      for (int i0 = threadIdx.x; i0 < S4; i0 += 512) {
        float4 dd[2], vv[2], zq[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = i0 + k * 256;
          ok[k] = i < S4;
          if (ok[k]) {
            dd[k] = ld4(dzp, i, nt);
            vv[k] = ld4(yp, i, nt);
            if (relu && zp) zq[k] = reinterpret_cast<const float4*>(zp)[i];
          }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!ok[k]) continue;
          const float4 d = dd[k], v = vv[k];
          float g0 = d.x, g1 = d.y, g2 = d.z, g3 = d.w;
          if (relu) {
            if (zp) {
              const float4 zz = zq[k];
              g0 = zz.x > 0.f ? g0 : 0.f; g1 = zz.y > 0.f ? g1 : 0.f;
              g2 = zz.z > 0.f ? g2 : 0.f; g3 = zz.w > 0.f ? g3 : 0.f;
            } else {
              g0 = fmaf(v.x, sc, sf) > 0.f ? g0 : 0.f; g1 = fmaf(v.y, sc, sf) > 0.f ? g1 : 0.f;
              g2 = fmaf(v.z, sc, sf) > 0.f ? g2 : 0.f; g3 = fmaf(v.w, sc, sf) > 0.f ? g3 : 0.f;
            }
          }
          ag += (g0 + g1) + (g2 + g3);
          agx += g0 * ((v.x - mu) * is) + g1 * ((v.y - mu) * is) + g2 * ((v.z - mu) * is) +
                 g3 * ((v.w - mu) * is);
        }
      }
    } else {
      for (int i = threadIdx.x; i < S; i += 256) {
        float g = dzp[i];
        const float v = yp[i];
        if (relu) g = (zp ? zp[i] > 0.f : fmaf(v, sc, sf) > 0.f) ? g : 0.f;
        ag += g;
        agx += g * ((v - mu) * is);
      }
    }
    sg += (double)ag;
    sgx += (double)agx;
  }
  sg = block256_sum_d(sg, red);
  sgx = block256_sum_d(sgx, red);
  if (threadIdx.x == 0) {
    // one partial per (channel, sample group): no atomics, no zero-fill, fixed order
    sums[((long)c * gridDim.y + blockIdx.y) * 2] = sg;
    sums[((long)c * gridDim.y + blockIdx.y) * 2 + 1] = sgx;
  }
}

// The reduction done elsewhere: the data gradient(s) that wrote dz left (sum g, sum g*xhat) per channel
// and output tile in float ([2][C][ntiles], coclr_conv_call.bwd_y); fold them in fp64, fixed order, into
// the one-group form the apply pass reads.
__global__ void __launch_bounds__(256)
bn_bwd_fold_kernel(const float* __restrict__ p0, int n0, const float* __restrict__ p1, int n1, int C,
                   double* sums) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double sg = 0.0, sgx = 0.0;
  for (int i = threadIdx.x; i < n0; i += 256) {
    sg += (double)p0[(long)c * n0 + i];
    sgx += (double)p0[((long)C + c) * n0 + i];
  }
  if (p1)
    for (int i = threadIdx.x; i < n1; i += 256) {
      sg += (double)p1[(long)c * n1 + i];
      sgx += (double)p1[((long)C + c) * n1 + i];
    }
  sg = block256_sum_d(sg, red);
  sgx = block256_sum_d(sgx, red);
  if (threadIdx.x == 0) { sums[(long)c * 2] = sg; sums[(long)c * 2 + 1] = sgx; }
}

// Apply pass: folds the per-group partial sums of its channel (fp64), derives
//   training: dy = A*g + Bc*y + D  with  A = scale, Bc = -scale*invstd*mgx,
//             D = scale*(mean*invstd*mgx - mg);  eval: dy = scale*g
// and the block of sample 0 also emits dgamma / dbeta.
template <bool VEC>
__global__ void __launch_bounds__(256)
bn_act_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                        const float* __restrict__ z, const float* __restrict__ scale,
                        const float* __restrict__ shift, const float* __restrict__ mean,
                        const float* __restrict__ invstd, const double* __restrict__ sums,
                        int groups, double count, int training, float* dgamma, float* dbeta,
                        float* dy, float* dres, int N, int C, int S, long dz_nstride,
                        long y_nstride, long dy_nstride, long z_nstride, long dres_nstride,
                        int relu_nt, int dres_accumulate) {
  __shared__ double tot[2];
  const int relu = relu_nt & 1;
  const bool nt = (relu_nt & 2) != 0;
  // blockIdx.y = (sample group, channel); the partial sums of the channel are folded once per
  // block, then the block walks its samples
  const int c = blockIdx.y % C, sgrp = blockIdx.y / C, nsg = gridDim.y / C;
  if (threadIdx.x < 64) {
    double a0 = 0.0, a1 = 0.0;
    for (int g = threadIdx.x; g < groups; g += 64) {
      a0 += sums[((long)c * groups + g) * 2];
      a1 += sums[((long)c * groups + g) * 2 + 1];
    }
    a0 = wave_sum_d(a0);
    a1 = wave_sum_d(a1);
    if (threadIdx.x == 0) { tot[0] = a0; tot[1] = a1; }
  }
  __syncthreads();
  const double sg = tot[0], sgx = tot[1];
  const float sc = scale[c], sf = shift[c];
  float A = sc, B = 0.f, D = 0.f;
  if (training) {
    const float mg = (float)(sg / count), mgx = (float)(sgx / count);
    B = -sc * invstd[c] * mgx;
    D = sc * (mean[c] * invstd[c] * mgx - mg);
  }
  if (sgrp == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    if (dgamma) dgamma[c] = (float)sgx;
    if (dbeta) dbeta[c] = (float)sg;
  }
  for (int n = sgrp; n < N; n += nsg) {
    const float* dzp = dz + (long)n * dz_nstride + (long)c * S;
    const float* yp = y + (long)n * y_nstride + (long)c * S;
    const float* zp = z ? z + (long)n * z_nstride + (long)c * S : nullptr;
    float* dyp = dy + (long)n * dy_nstride + (long)c * S;
    float* drp = dres ? dres + (long)n * dres_nstride + (long)c * S : nullptr;
    if (VEC) {
      const int S4 = S >> 2, stride = gridDim.x * 256;
      // two (dz, y) pairs in flight per thread and trip
      for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < S4; i0 += 2 * stride) {
        float4 dd[2], vv[2], zq[2];
        bool ok[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = i0 + k * stride;
          ok[k] = i < S4;
          if (ok[k]) {
            dd[k] = ld4(dzp, i, nt);
            vv[k] = ld4(yp, i, nt);
            if (relu && zp) zq[k] = reinterpret_cast<const float4*>(zp)[i];
          }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          if (!ok[k]) continue;
          const int i = i0 + k * stride;
          const float4 v = vv[k];
          float4 g = dd[k];
          if (relu) {
            if (zp) {
              const float4 zz = zq[k];
              g.x = zz.x > 0.f ? g.x : 0.f; g.y = zz.y > 0.f ? g.y : 0.f;
              g.z = zz.z > 0.f ? g.z : 0.f; g.w = zz.w > 0.f ? g.w : 0.f;
            } else {
              g.x = fmaf(v.x, sc, sf) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, sc, sf) > 0.f ? g.y : 0.f;
              g.z = fmaf(v.z, sc, sf) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, sc, sf) > 0.f ? g.w : 0.f;
            }
          }
          if (drp) {
            float4 r = g;
            if (dres_accumulate) {
              const float4 o = reinterpret_cast<const float4*>(drp)[i];
              r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
            }
            reinterpret_cast<float4*>(drp)[i] = r;
          }
          float4 o;
          o.x = fmaf(A, g.x, fmaf(B, v.x, D)); o.y = fmaf(A, g.y, fmaf(B, v.y, D));
          o.z = fmaf(A, g.z, fmaf(B, v.z, D)); o.w = fmaf(A, g.w, fmaf(B, v.w, D));
          st4(dyp, i, o, nt);
        }
      }
    } else {
      for (int i = blockIdx.x * 256 + threadIdx.x; i < S; i += gridDim.x * 256) {
        float g = dzp[i];
        const float v = yp[i];
        if (relu) g = (zp ? zp[i] > 0.f : fmaf(v, sc, sf) > 0.f) ? g : 0.f;
        if (drp) drp[i] = dres_accumulate ? drp[i] + g : g;
        dyp[i] = fmaf(A, g, fmaf(B, v, D));
      }
    }
  }
}

// The apply pass's coefficients WITHOUT the apply pass: coef[5][C] = A, B, D (see bn_act_bwd_apply_kernel:
// same fold, same expressions, so a consumer that forms fmaf(A, g, fmaf(B, y, D)) itself reproduces its dy bit
// for bit), scale, shift; plus dgamma / dbeta.  For units whose d(conv output) has one reader that can apply
// it while it loads (coclr_conv3d_wgrad_bn).
__global__ void __launch_bounds__(64)
bn_bwd_coeffs_kernel(const float* __restrict__ scale, const float* __restrict__ shift,
                     const float* __restrict__ mean, const float* __restrict__ invstd,
                     const double* __restrict__ sums, int groups, double count, int training, int C,
                     float* __restrict__ coef, float* dgamma, float* dbeta) {
  const int c = blockIdx.x;
  double a0 = 0.0, a1 = 0.0;
  for (int g = threadIdx.x; g < groups; g += 64) {
    a0 += sums[((long)c * groups + g) * 2];
    a1 += sums[((long)c * groups + g) * 2 + 1];
  }
  a0 = wave_sum_d(a0);
  a1 = wave_sum_d(a1);
  if (threadIdx.x != 0) return;
  const double sg = a0, sgx = a1;
  const float sc = scale[c];
  float A = sc, B = 0.f, D = 0.f;
  if (training) {
    const float mg = (float)(sg / count), mgx = (float)(sgx / count);
    B = -sc * invstd[c] * mgx;
    D = sc * (mean[c] * invstd[c] * mgx - mg);
  }
  coef[c] = A; coef[C + c] = B; coef[2 * C + c] = D; coef[3 * C + c] = sc; coef[4 * C + c] = shift[c];
  if (dgamma) dgamma[c] = (float)sgx;
  if (dbeta) dbeta[c] = (float)sg;
}

// ---- small layers: one launch per BatchNorm unit ---------------------------------------------
// In the last two stages of S3D a channel holds N*S = 2048..16384 values (4x4x4 / 8x8x8 maps): the
// two-kernel forms above are launch-bound there (profiles/r01_e_layers.txt: 0.5-3.6 TB/s).  One
// workgroup per channel does everything for its channel in one launch.

// forward: fold the conv's partial sums, coefficients + running statistics, then z = act(y*sc+sf)
template <bool VEC>
__device__ __forceinline__ void
bn_fwd_fused_body(const int c, const float* __restrict__ sum, const float* __restrict__ sumsq, int ntiles,
                  double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                  float* running_mean, float* running_var, int64_t* num_batches_tracked,
                  float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out,
                  float* shift_out, const float* __restrict__ y, float* z, int N, int S,
                  long y_nstride, long z_nstride, int relu) {
  __shared__ double red[4];
  __shared__ float coef[2];
  const float* ps = sum + (long)c * ntiles;
  const float* pq = sumsq + (long)c * ntiles;
  double s = 0.0, q = 0.0;
  for (int i = threadIdx.x; i < ntiles; i += 256) { s += (double)ps[i]; q += (double)pq[i]; }
  s = block256_sum_d(s, red);
  q = block256_sum_d(q, red);
  if (threadIdx.x == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd, sf = b - (float)mean * sc;
    mean_out[c] = (float)mean;
    invstd_out[c] = invstd;
    scale_out[c] = sc;
    shift_out[c] = sf;
    coef[0] = sc; coef[1] = sf;
    if (running_mean) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
  }
  __syncthreads();
  const float sc = coef[0], sf = coef[1];
  if (VEC) {
    // One workgroup walks its whole channel (these launches have fewer workgroups than the chip has CUs):
    // four independent loads per thread and trip, or every trip costs a full memory round trip
    const int S4 = S >> 2, total = N * S4;
    for (int e0 = threadIdx.x; e0 < total; e0 += 1024) {
      float4 v[4];
      long zo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        zo[k] = -1;
        if (e < total) {
          const int n = e / S4, i = e - n * S4;
          v[k] = reinterpret_cast<const float4*>(y + (long)n * y_nstride + (long)c * S)[i];
          zo[k] = (long)n * z_nstride + (long)c * S + 4 * i;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (zo[k] < 0) continue;
        float4 w = v[k];
        w.x = fmaf(w.x, sc, sf); w.y = fmaf(w.y, sc, sf); w.z = fmaf(w.z, sc, sf); w.w = fmaf(w.w, sc, sf);
        if (relu) {
          w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f);
        }
        *reinterpret_cast<float4*>(z + zo[k]) = w;
      }
    }
  } else {
    const int total = N * S;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int n = e / S, i = e - n * S;
      float v = fmaf(y[(long)n * y_nstride + (long)c * S + i], sc, sf);
      if (relu) v = fmaxf(v, 0.f);
      z[(long)n * z_nstride + (long)c * S + i] = v;
    }
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256)
bn_fwd_fused_kernel(const float* __restrict__ sum, const float* __restrict__ sumsq, int ntiles,
                    double count, const float* __restrict__ gamma, const float* __restrict__ beta,
                    float* running_mean, float* running_var, int64_t* num_batches_tracked,
                    float momentum, float eps, float* mean_out, float* invstd_out, float* scale_out,
                    float* shift_out, const float* __restrict__ y, float* z, int N, int S,
                    long y_nstride, long z_nstride, int relu) {
  bn_fwd_fused_body<VEC>((int)blockIdx.x, sum, sumsq, ntiles, count, gamma, beta, running_mean, running_var,
                         num_batches_tracked, momentum, eps, mean_out, invstd_out, scale_out, shift_out, y,
                         z, N, S, y_nstride, z_nstride, relu);
}

// Up to four BatchNorm units in ONE launch (the three 1x1x1 heads of an inception block; the two
// separable branches side by side): the grid is the units' channels back to back, a workgroup finds its
// unit by its index.  Same body, same arithmetic, same results as the single-unit launches.
struct BnFwdUnit {
  const float* sum; const float* sumsq; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* nbt;
  float* mean; float* invstd; float* scale; float* shift;
  const float* y; float* z;
  double count;
  long y_nstride, z_nstride;
  int C, ntiles, N, S, relu;
  float momentum, eps;
};
struct BnFwdTable { BnFwdUnit u[4]; int n; };

template <bool VEC>
__global__ void __launch_bounds__(256)
bn_fwd_fused_multi_kernel(const BnFwdTable t) {
  int c = (int)blockIdx.x;
#define COCLR_BN_FWD_UNIT(k)                                                                              \
  if (k < t.n) {                                                                                         \
    if (c < t.u[k].C) {                                                                                  \
      bn_fwd_fused_body<VEC>(c, t.u[k].sum, t.u[k].sumsq, t.u[k].ntiles, t.u[k].count, t.u[k].gamma,     \
                             t.u[k].beta, t.u[k].running_mean, t.u[k].running_var, t.u[k].nbt,           \
                             t.u[k].momentum, t.u[k].eps, t.u[k].mean, t.u[k].invstd, t.u[k].scale,      \
                             t.u[k].shift, t.u[k].y, t.u[k].z, t.u[k].N, t.u[k].S, t.u[k].y_nstride,     \
                             t.u[k].z_nstride, t.u[k].relu);                                             \
      return;                                                                                            \
    }                                                                                                    \
    c -= t.u[k].C;                                                                                       \
  }
  COCLR_BN_FWD_UNIT(0)
  COCLR_BN_FWD_UNIT(1)
  COCLR_BN_FWD_UNIT(2)
  COCLR_BN_FWD_UNIT(3)
#undef COCLR_BN_FWD_UNIT
}

// backward: sums of g and g*xhat over the channel, then dy = A*g + B*y + D (training) / scale*g
// (eval); the second read of dz / y comes out of L2 (<= 128 KB per workgroup).
template <bool VEC>
__device__ __forceinline__ void
bn_bwd_fused_body(const int c, const float* __restrict__ dz, const float* __restrict__ y,
                  const float* __restrict__ scale, const float* __restrict__ shift,
                  const float* __restrict__ mean, const float* __restrict__ invstd, int training,
                  float* dgamma, float* dbeta, float* dy, int N, int S, long dz_nstride,
                  long y_nstride, long dy_nstride, int relu) {
  __shared__ double red[4];
  const float sc = scale[c], sf = shift[c], mu = mean[c], is = invstd[c];
  double sg = 0.0, sgx = 0.0;
  if (VEC) {
    const int S4 = S >> 2, total = N * S4;
    float ag = 0.f, agx = 0.f;
    // four independent (dz, y) loads per thread and trip (see bn_fwd_fused_kernel); the per-thread
    // summation order is the one-element-per-trip loop's
    for (int e0 = threadIdx.x; e0 < total; e0 += 1024) {
      float4 dd[4], vv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        if (e < total) {
          const int n = e / S4, i = e - n * S4;
          dd[k] = reinterpret_cast<const float4*>(dz + (long)n * dz_nstride + (long)c * S)[i];
          vv[k] = reinterpret_cast<const float4*>(y + (long)n * y_nstride + (long)c * S)[i];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        if (e >= total) break;
        const float4 d = dd[k], v = vv[k];
        float g0 = d.x, g1 = d.y, g2 = d.z, g3 = d.w;
        if (relu) {
          g0 = fmaf(v.x, sc, sf) > 0.f ? g0 : 0.f; g1 = fmaf(v.y, sc, sf) > 0.f ? g1 : 0.f;
          g2 = fmaf(v.z, sc, sf) > 0.f ? g2 : 0.f; g3 = fmaf(v.w, sc, sf) > 0.f ? g3 : 0.f;
        }
        ag += (g0 + g1) + (g2 + g3);
        agx += g0 * ((v.x - mu) * is) + g1 * ((v.y - mu) * is) + g2 * ((v.z - mu) * is) +
               g3 * ((v.w - mu) * is);
        if ((e & 0xfff) == 0xfff) { sg += (double)ag; sgx += (double)agx; ag = agx = 0.f; }
      }
    }
    sg += (double)ag; sgx += (double)agx;
  } else {
    const int total = N * S;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int n = e / S, i = e - n * S;
      float g = dz[(long)n * dz_nstride + (long)c * S + i];
      const float v = y[(long)n * y_nstride + (long)c * S + i];
      if (relu) g = fmaf(v, sc, sf) > 0.f ? g : 0.f;
      sg += (double)g;
      sgx += (double)(g * ((v - mu) * is));
    }
  }
  sg = block256_sum_d(sg, red);
  sgx = block256_sum_d(sgx, red);
  float A = sc, B = 0.f, D = 0.f;
  if (training) {
    const double count = (double)N * (double)S;
    const float mg = (float)(sg / count), mgx = (float)(sgx / count);
    B = -sc * is * mgx;
    D = sc * (mu * is * mgx - mg);
  }
  if (threadIdx.x == 0) {
    if (dgamma) dgamma[c] = (float)sgx;
    if (dbeta) dbeta[c] = (float)sg;
  }
  if (VEC) {
    const int S4 = S >> 2, total = N * S4;
    for (int e0 = threadIdx.x; e0 < total; e0 += 1024) {
      float4 dd[4], vv[4];
      long yo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        yo[k] = -1;
        if (e < total) {
          const int n = e / S4, i = e - n * S4;
          dd[k] = reinterpret_cast<const float4*>(dz + (long)n * dz_nstride + (long)c * S)[i];
          vv[k] = reinterpret_cast<const float4*>(y + (long)n * y_nstride + (long)c * S)[i];
          yo[k] = (long)n * dy_nstride + (long)c * S + 4 * i;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (yo[k] < 0) continue;
        const float4 v = vv[k];
        float4 g = dd[k];
        if (relu) {
          g.x = fmaf(v.x, sc, sf) > 0.f ? g.x : 0.f; g.y = fmaf(v.y, sc, sf) > 0.f ? g.y : 0.f;
          g.z = fmaf(v.z, sc, sf) > 0.f ? g.z : 0.f; g.w = fmaf(v.w, sc, sf) > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = fmaf(A, g.x, fmaf(B, v.x, D)); o.y = fmaf(A, g.y, fmaf(B, v.y, D));
        o.z = fmaf(A, g.z, fmaf(B, v.z, D)); o.w = fmaf(A, g.w, fmaf(B, v.w, D));
        *reinterpret_cast<float4*>(dy + yo[k]) = o;
      }
    }
  } else {
    const int total = N * S;
    for (int e = threadIdx.x; e < total; e += 256) {
      const int n = e / S, i = e - n * S;
      float g = dz[(long)n * dz_nstride + (long)c * S + i];
      const float v = y[(long)n * y_nstride + (long)c * S + i];
      if (relu) g = fmaf(v, sc, sf) > 0.f ? g : 0.f;
      dy[(long)n * dy_nstride + (long)c * S + i] = fmaf(A, g, fmaf(B, v, D));
    }
  }
}

template <bool VEC>
__global__ void __launch_bounds__(256)
bn_bwd_fused_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                    const float* __restrict__ scale, const float* __restrict__ shift,
                    const float* __restrict__ mean, const float* __restrict__ invstd, int training,
                    float* dgamma, float* dbeta, float* dy, int N, int S, long dz_nstride,
                    long y_nstride, long dy_nstride, int relu) {
  bn_bwd_fused_body<VEC>((int)blockIdx.x, dz, y, scale, shift, mean, invstd, training, dgamma, dbeta, dy, N, S,
                         dz_nstride, y_nstride, dy_nstride, relu);
}

struct BnBwdUnit {
  const float* dz; const float* y; const float* scale; const float* shift; const float* mean;
  const float* invstd;
  float* dgamma; float* dbeta; float* dy;
  long dz_nstride, y_nstride, dy_nstride;
  int C, N, S, relu, training;
};
struct BnBwdTable { BnBwdUnit u[4]; int n; };

template <bool VEC>
__global__ void __launch_bounds__(256)
bn_bwd_fused_multi_kernel(const BnBwdTable t) {
  int c = (int)blockIdx.x;
#define COCLR_BN_BWD_UNIT(k)                                                                              \
  if (k < t.n) {                                                                                         \
    if (c < t.u[k].C) {                                                                                  \
      bn_bwd_fused_body<VEC>(c, t.u[k].dz, t.u[k].y, t.u[k].scale, t.u[k].shift, t.u[k].mean,            \
                             t.u[k].invstd, t.u[k].training, t.u[k].dgamma, t.u[k].dbeta, t.u[k].dy,     \
                             t.u[k].N, t.u[k].S, t.u[k].dz_nstride, t.u[k].y_nstride,                    \
                             t.u[k].dy_nstride, t.u[k].relu);                                            \
      return;                                                                                            \
    }                                                                                                    \
    c -= t.u[k].C;                                                                                       \
  }
  COCLR_BN_BWD_UNIT(0)
  COCLR_BN_BWD_UNIT(1)
  COCLR_BN_BWD_UNIT(2)
  COCLR_BN_BWD_UNIT(3)
#undef COCLR_BN_BWD_UNIT
}

constexpr long kSmallChannel = 32768;     // N*S per channel up to which one workgroup per channel wins

// Grid for the streaming kernels: x = chunks of one plane, y = (sample groups) x C.
// Every block should see >= ~4096 elements: big planes get one block (or several) per
// (n, c) plane, the 64..512-element planes of the last stages several samples per block.
inline dim3 plane_grid(int N, int C, int S) {
  int gx = cdiv(S >> 2 ? S >> 2 : S, 256 * 4);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  int nsg = N;
  if (S < 4096) {
    const long per = 4096 / S;            // samples per block
    nsg = cdiv(N, per);
  }
  if ((long)nsg * C > 65535) nsg = 65535 / C;
  if (nsg < 1) nsg = 1;
  // keep total blocks bounded for very large tensors
  while ((long)gx * nsg * C > 262144 && gx > 1) gx >>= 1;
  return dim3(gx, nsg * C);
}

// sample groups of the backward reduce (one fp64 partial pair per (channel, group))
inline int reduce_groups(int N, int S) {
  if (S >= 4096) return N;
  int g = cdiv(N, 4096 / S);
  return g < 1 ? 1 : g;
}

}  // namespace

extern "C" int coclr_bn_finalize(const float* sum, const float* sumsq, int C, int ntiles,
                                 double count, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, float momentum, float eps, float* mean,
                                 float* invstd, float* scale, float* shift, void* stream) {
  if (C <= 0 || ntiles <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, sum, sumsq, C,
                     ntiles, count, gamma, beta, running_mean, running_var, num_batches_tracked,
                     momentum, eps, mean, invstd, scale, shift);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_finalize_apply(const float* sum, const float* sumsq, int C, int ntiles,
                                       double count, const float* gamma, const float* beta,
                                       float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float momentum, float eps,
                                       float* mean, float* invstd, float* scale, float* shift,
                                       const float* y, float* z, int N, int64_t S, int64_t y_nstride,
                                       int64_t z_nstride, int relu, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0 || ntiles <= 0 || N <= 0 || S <= 0 || !y || !z) return COCLR_EINVAL;
  if ((long)N * S > kSmallChannel) {
    // large layers: statistics in one launch, the streaming apply pass in another
    int rc = coclr_bn_finalize(sum, sumsq, C, ntiles, count, gamma, beta, running_mean, running_var,
                               num_batches_tracked, momentum, eps, mean, invstd, scale, shift, stream_);
    if (rc) return rc;
    return coclr_bn_act_apply(y, scale, shift, nullptr, z, N, C, S, y_nstride, z_nstride, 0, relu,
                              stream_);
  }
  const bool vec = (S % 4 == 0) && (y_nstride % 4 == 0) && (z_nstride % 4 == 0);
  if (vec)
    hipLaunchKernelGGL(bn_fwd_fused_kernel<true>, dim3(C), dim3(256), 0, stream, sum, sumsq, ntiles,
                       count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                       eps, mean, invstd, scale, shift, y, z, N, (int)S, (long)y_nstride,
                       (long)z_nstride, relu);
  else
    hipLaunchKernelGGL(bn_fwd_fused_kernel<false>, dim3(C), dim3(256), 0, stream, sum, sumsq, ntiles,
                       count, gamma, beta, running_mean, running_var, num_batches_tracked, momentum,
                       eps, mean, invstd, scale, shift, y, z, N, (int)S, (long)y_nstride,
                       (long)z_nstride, relu);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_finalize_apply_multi(const coclr_bn_fwd_call* calls, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!calls || n < 1) return COCLR_EINVAL;
  const char* env = getenv("COCLR_PAIR");
  const bool fuse = !(env && env[0] == '0');
  int i = 0;
  while (i < n) {
    // a run of up to four small units (one workgroup per channel) with the same vector width
    BnFwdTable t;
    t.n = 0;
    long blocks = 0;
    bool vec0 = false;
    int j = i;
    for (; j < n && t.n < 4; ++j) {
      const coclr_bn_fwd_call& c = calls[j];
      if (c.C <= 0 || c.ntiles <= 0 || c.N <= 0 || c.S <= 0 || !c.y || !c.z) return COCLR_EINVAL;
      const bool small = (long)c.N * c.S <= kSmallChannel;
      const bool vec = (c.S % 4 == 0) && (c.y_nstride % 4 == 0) && (c.z_nstride % 4 == 0);
      if (!small || !fuse) break;
      if (t.n == 0) vec0 = vec;
      else if (vec != vec0) break;
      BnFwdUnit& u = t.u[t.n++];
      u.sum = c.sum; u.sumsq = c.sumsq; u.gamma = c.gamma; u.beta = c.beta;
      u.running_mean = c.running_mean; u.running_var = c.running_var; u.nbt = c.num_batches_tracked;
      u.mean = c.mean; u.invstd = c.invstd; u.scale = c.scale; u.shift = c.shift;
      u.y = c.y; u.z = c.z; u.count = c.count;
      u.y_nstride = (long)c.y_nstride; u.z_nstride = (long)c.z_nstride;
      u.C = c.C; u.ntiles = c.ntiles; u.N = c.N; u.S = (int)c.S; u.relu = c.relu;
      u.momentum = c.momentum; u.eps = c.eps;
      blocks += c.C;
    }
    if (t.n >= 2) {
      if (vec0)
        hipLaunchKernelGGL(bn_fwd_fused_multi_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, t);
      else
        hipLaunchKernelGGL(bn_fwd_fused_multi_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, t);
      COCLR_LAUNCH_CHECK();
      i = j;
      continue;
    }
    const coclr_bn_fwd_call& c = calls[i];
    int rc = coclr_bn_finalize_apply(c.sum, c.sumsq, c.C, c.ntiles, c.count, c.gamma, c.beta,
                                     c.running_mean, c.running_var, c.num_batches_tracked, c.momentum,
                                     c.eps, c.mean, c.invstd, c.scale, c.shift, c.y, c.z, c.N, c.S,
                                     c.y_nstride, c.z_nstride, c.relu, stream_);
    if (rc) return rc;
    ++i;
  }
  return 0;
}

extern "C" int coclr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                                    const float* running_var, float eps, int C, float* mean,
                                    float* invstd, float* scale, float* shift, void* stream) {
  if (C <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, running_mean, running_var, eps, C, mean, invstd, scale, shift);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_act_apply(const float* y, const float* scale, const float* shift,
                                  const float* residual, float* z, int N, int C, int64_t S,
                                  int64_t y_nstride, int64_t z_nstride, int64_t res_nstride,
                                  int relu, void* stream) {
  if (N <= 0 || C <= 0 || S <= 0) return COCLR_EINVAL;
  const bool vec = (S % 4 == 0) && (y_nstride % 4 == 0) && (z_nstride % 4 == 0) &&
                   (!residual || res_nstride % 4 == 0);
  dim3 grid = plane_grid(N, C, (int)S);
  // Non-temporal loads / stores in the streaming BatchNorm passes: they move tensors of up to 2 GB through a chip
  // whose other streams run MFMA-bound kernels that live on what they keep re-reading from L2 / the 256 MB
  // infinity cache (weights, stencil windows, split-K partials).  Measured inside the step, same box, alternating
  // x4 (profiles/r06_bn_nontemporal_ab.txt): +1.3 % with every such pass non-temporal, +0.9 % with only the
  // tensors above 200 MB.  COCLR_BN_NT_MB=<MB> sets the size from which a pass is non-temporal, -1 switches it off.
  static const long nt_bytes = bn_nt_bytes();
  if (vec && nt_bytes >= 0 && (long)N * C * S * 4 >= nt_bytes)
    hipLaunchKernelGGL((bn_act_apply_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, y, scale,
                       shift, residual, z, N, C, (int)S, (long)y_nstride, (long)z_nstride,
                       (long)res_nstride, relu);
  else if (vec)
    hipLaunchKernelGGL(bn_act_apply_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, y, scale,
                       shift, residual, z, N, C, (int)S, (long)y_nstride, (long)z_nstride,
                       (long)res_nstride, relu);
  else
    hipLaunchKernelGGL(bn_act_apply_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, y, scale,
                       shift, residual, z, N, C, (int)S, (long)y_nstride, (long)z_nstride,
                       (long)res_nstride, relu);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_backward_workspace(int N, int C, int64_t* doubles) {
  if (N <= 0 || C <= 0) return COCLR_EINVAL;
  *doubles = (int64_t)2 * C * N;
  return 0;
}

extern "C" int coclr_bn_act_backward(const float* dz, const float* y, const float* z,
                                     const float* scale, const float* shift, const float* mean,
                                     const float* invstd, double* sums_ws, float* dy, float* dres,
                                     float* dgamma, float* dbeta, int N, int C, int64_t S,
                                     int64_t dz_nstride, int64_t y_nstride, int64_t dy_nstride,
                                     int64_t z_nstride, int64_t dres_nstride, int relu,
                                     int training, int dres_accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N <= 0 || C <= 0 || S <= 0) return COCLR_EINVAL;
  const bool vec = (S % 4 == 0) && (dz_nstride % 4 == 0) && (y_nstride % 4 == 0) &&
                   (dy_nstride % 4 == 0) && (!z || z_nstride % 4 == 0) &&
                   (!dres || dres_nstride % 4 == 0);
  if (!z && !dres && (long)N * S <= kSmallChannel) {
    // small layers: one workgroup per channel, one launch (the workspace is not used)
    if (vec)
      hipLaunchKernelGGL(bn_bwd_fused_kernel<true>, dim3(C), dim3(256), 0, stream, dz, y, scale, shift,
                         mean, invstd, training, dgamma, dbeta, dy, N, (int)S, (long)dz_nstride,
                         (long)y_nstride, (long)dy_nstride, relu);
    else
      hipLaunchKernelGGL(bn_bwd_fused_kernel<false>, dim3(C), dim3(256), 0, stream, dz, y, scale,
                         shift, mean, invstd, training, dgamma, dbeta, dy, N, (int)S,
                         (long)dz_nstride, (long)y_nstride, (long)dy_nstride, relu);
    COCLR_LAUNCH_CHECK();
    return 0;
  }
  static const long nt_bytes_b = bn_nt_bytes();          // see coclr_bn_act_apply
  relu = relu ? 1 : 0;
  if (nt_bytes_b >= 0 && vec && (long)N * C * S * 4 >= nt_bytes_b) relu |= 2;   // bit 1: non-temporal passes
  // pass 1: per (channel, sample group) partial sums of g and g*xhat
  const int groups = reduce_groups(N, (int)S);
  dim3 rgrid(C, groups);
  if (vec)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<true>, rgrid, dim3(256), 0, stream, dz, y, z, scale,
                       shift, mean, invstd, sums_ws, N, C, (int)S, (long)dz_nstride,
                       (long)y_nstride, (long)z_nstride, relu);
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<false>, rgrid, dim3(256), 0, stream, dz, y, z,
                       scale, shift, mean, invstd, sums_ws, N, C, (int)S, (long)dz_nstride,
                       (long)y_nstride, (long)z_nstride, relu);
  COCLR_LAUNCH_CHECK();
  // pass 2: fold the partials, coefficients, dy (+ dres), dgamma / dbeta
  const double count = (double)N * (double)S;
  dim3 grid = plane_grid(N, C, (int)S);
  if (vec)
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, grid, dim3(256), 0, stream, dz, y, z, scale,
                       shift, mean, invstd, sums_ws, groups, count, training, dgamma, dbeta, dy, dres,
                       N, C, (int)S, (long)dz_nstride, (long)y_nstride, (long)dy_nstride,
                       (long)z_nstride, (long)dres_nstride, relu, dres_accumulate);
  else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, grid, dim3(256), 0, stream, dz, y, z, scale,
                       shift, mean, invstd, sums_ws, groups, count, training, dgamma, dbeta, dy, dres,
                       N, C, (int)S, (long)dz_nstride, (long)y_nstride, (long)dy_nstride,
                       (long)z_nstride, (long)dres_nstride, relu, dres_accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_act_backward_coeffs(const float* dz, const float* y, const float* scale,
                                            const float* shift, const float* mean, const float* invstd,
                                            double* sums_ws, float* coef, float* dgamma, float* dbeta,
                                            int N, int C, int64_t S, int64_t dz_nstride, int64_t y_nstride,
                                            int relu, int training, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (N <= 0 || C <= 0 || S <= 0 || !coef || !sums_ws) return COCLR_EINVAL;
  const bool vec = (S % 4 == 0) && (dz_nstride % 4 == 0) && (y_nstride % 4 == 0);
  static const long nt_bytes_c = bn_nt_bytes();          // see coclr_bn_act_apply
  relu = relu ? 1 : 0;
  if (nt_bytes_c >= 0 && vec && (long)N * C * S * 4 >= nt_bytes_c) relu |= 2;
  const int groups = reduce_groups(N, (int)S);
  dim3 rgrid(C, groups);
  const float* noz = nullptr;
  if (vec)
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<true>, rgrid, dim3(256), 0, stream, dz, y, noz, scale,
                       shift, mean, invstd, sums_ws, N, C, (int)S, (long)dz_nstride, (long)y_nstride, 0L,
                       relu);
  else
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel<false>, rgrid, dim3(256), 0, stream, dz, y, noz, scale,
                       shift, mean, invstd, sums_ws, N, C, (int)S, (long)dz_nstride, (long)y_nstride, 0L,
                       relu);
  COCLR_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(64), 0, stream, scale, shift, mean, invstd, sums_ws,
                     groups, (double)N * (double)S, training, C, coef, dgamma, dbeta);
  COCLR_LAUNCH_CHECK();
  return 0;
}

namespace {
// BatchNorm(+ReLU) backward of a unit whose sums the producing data gradient already formed: fold + apply
int bn_backward_from_partials(const coclr_bn_bwd_call& c, hipStream_t stream) {
  if (!c.sums_ws || c.part_ntiles[0] <= 0 || (c.part[1] && c.part_ntiles[1] <= 0)) return COCLR_EINVAL;
  hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3(c.C), dim3(256), 0, stream, c.part[0], c.part_ntiles[0],
                     c.part[1], c.part[1] ? c.part_ntiles[1] : 0, c.C, c.sums_ws);
  COCLR_LAUNCH_CHECK();
  const bool vec = (c.S % 4 == 0) && (c.dz_nstride % 4 == 0) && (c.y_nstride % 4 == 0) &&
                   (c.dy_nstride % 4 == 0);
  const double count = (double)c.N * (double)c.S;
  dim3 grid = plane_grid(c.N, c.C, (int)c.S);
  const float* nof = nullptr;
  float* nod = nullptr;
  if (vec)
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<true>, grid, dim3(256), 0, stream, c.dz, c.y, nof, c.scale,
                       c.shift, c.mean, c.invstd, c.sums_ws, 1, count, c.training, c.dgamma, c.dbeta, c.dy, nod,
                       c.N, c.C, (int)c.S, (long)c.dz_nstride, (long)c.y_nstride, (long)c.dy_nstride, 0L, 0L,
                       c.relu, 0);
  else
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel<false>, grid, dim3(256), 0, stream, c.dz, c.y, nof, c.scale,
                       c.shift, c.mean, c.invstd, c.sums_ws, 1, count, c.training, c.dgamma, c.dbeta, c.dy, nod,
                       c.N, c.C, (int)c.S, (long)c.dz_nstride, (long)c.y_nstride, (long)c.dy_nstride, 0L, 0L,
                       c.relu, 0);
  COCLR_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int coclr_bn_act_backward_multi(const coclr_bn_bwd_call* calls, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!calls || n < 1) return COCLR_EINVAL;
  const char* env = getenv("COCLR_PAIR");
  const bool fuse = !(env && env[0] == '0');
  int i = 0;
  while (i < n) {
    BnBwdTable t;
    t.n = 0;
    long blocks = 0;
    bool vec0 = false;
    int j = i;
    for (; j < n && t.n < 4; ++j) {
      const coclr_bn_bwd_call& c = calls[j];
      if (c.N <= 0 || c.C <= 0 || c.S <= 0 || !c.dz || !c.y || !c.dy) return COCLR_EINVAL;
      const bool small = (long)c.N * c.S <= kSmallChannel;
      const bool vec = (c.S % 4 == 0) && (c.dz_nstride % 4 == 0) && (c.y_nstride % 4 == 0) &&
                       (c.dy_nstride % 4 == 0);
      if (!small || !fuse || c.part[0]) break;
      if (t.n == 0) vec0 = vec;
      else if (vec != vec0) break;
      BnBwdUnit& u = t.u[t.n++];
      u.dz = c.dz; u.y = c.y; u.scale = c.scale; u.shift = c.shift; u.mean = c.mean; u.invstd = c.invstd;
      u.dgamma = c.dgamma; u.dbeta = c.dbeta; u.dy = c.dy;
      u.dz_nstride = (long)c.dz_nstride; u.y_nstride = (long)c.y_nstride; u.dy_nstride = (long)c.dy_nstride;
      u.C = c.C; u.N = c.N; u.S = (int)c.S; u.relu = c.relu; u.training = c.training;
      blocks += c.C;
    }
    if (t.n >= 2) {
      if (vec0)
        hipLaunchKernelGGL(bn_bwd_fused_multi_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, t);
      else
        hipLaunchKernelGGL(bn_bwd_fused_multi_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, t);
      COCLR_LAUNCH_CHECK();
      i = j;
      continue;
    }
    const coclr_bn_bwd_call& c = calls[i];
    if (c.part[0]) {
      int rc = bn_backward_from_partials(c, stream);
      if (rc) return rc;
      ++i;
      continue;
    }
    int rc = coclr_bn_act_backward(c.dz, c.y, nullptr, c.scale, c.shift, c.mean, c.invstd, c.sums_ws, c.dy,
                                   nullptr, c.dgamma, c.dbeta, c.N, c.C, c.S, c.dz_nstride, c.y_nstride,
                                   c.dy_nstride, 0, 0, c.relu, c.training, 0, stream_);
    if (rc) return rc;
    ++i;
  }
  return 0;
}
