// MaxPool3d (floor mode, -inf padding, first-max-wins like ATen) with argmax
// indices, its gather-style backward, and the global average pool of the
// projection head.  Replaces ATen max_pool3d_with_indices(+_backward) used at
// backbone/s3dg.py:105,151,162,173,190 / backbone/resnet_2d3d.py:141 and
// adaptive_avg_pool3d((1,1,1)) at model/pretrain.py:51.
//
// HBM-bound stencil kernels: one lane per output (forward) / per input
// (backward, no atomics -> deterministic), w fastest so a wave touches
// contiguous rows.
#include "common.h"
#include "../../include/coclr_hip.h"
#include <math.h>

namespace {

struct PoolGeom {
  int N, C;
  int Ti, Hi, Wi, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  long x_nstride, y_nstride;   // floats between samples (channel slices allowed)
};

__global__ void __launch_bounds__(256)
maxpool3d_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                     const PoolGeom g) {
  const int So = g.To * g.Ho * g.Wo, Si = g.Ti * g.Hi * g.Wi;
  const int planes = g.N * g.C;
  for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const int n = pl / g.C, c = pl - n * g.C;
    const float* xp = x + (long)n * g.x_nstride + (long)c * Si;
    float* yp = y + (long)n * g.y_nstride + (long)c * So;
    int* ip = idx ? idx + (long)pl * So : nullptr;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < So; o += gridDim.x * 256) {
      const int ow = o % g.Wo;
      const int q = o / g.Wo;
      const int oh = q % g.Ho, ot = q / g.Ho;
      int t0 = ot * g.st - g.pt, h0 = oh * g.sh - g.ph, w0 = ow * g.sw - g.pw;
      const int t1 = min(t0 + g.kt, g.Ti), h1 = min(h0 + g.kh, g.Hi), w1 = min(w0 + g.kw, g.Wi);
      t0 = max(t0, 0); h0 = max(h0, 0); w0 = max(w0, 0);
      float best = -INFINITY;
      int bi = (t0 * g.Hi + h0) * g.Wi + w0;
      for (int t = t0; t < t1; ++t)
        for (int h = h0; h < h1; ++h)
          for (int w = w0; w < w1; ++w) {
            const int ii = (t * g.Hi + h) * g.Wi + w;
            const float v = xp[ii];
            if (v > best || v != v) { best = v; bi = ii; }
          }
      yp[o] = best;
      if (ip) ip[o] = bi;
    }
  }
}

// smallest o with o*s > a  (a = i + pad - k)
__device__ __forceinline__ int first_out(int a, int s) { return a < 0 ? 0 : a / s + 1; }

// dx[i] (+)= sum over outputs whose window holds i and whose argmax == i.
__global__ void __launch_bounds__(256)
maxpool3d_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* dx,
                     const PoolGeom g, long dy_nstride, long dx_nstride, int accumulate) {
  const int So = g.To * g.Ho * g.Wo, Si = g.Ti * g.Hi * g.Wi;
  const int planes = g.N * g.C;
  for (int pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const int n = pl / g.C, c = pl - n * g.C;
    const float* dyp = dy + (long)n * dy_nstride + (long)c * So;
    const int* ip = idx + (long)pl * So;
    float* dxp = dx + (long)n * dx_nstride + (long)c * Si;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Si; i += gridDim.x * 256) {
      const int iw = i % g.Wi;
      const int q = i / g.Wi;
      const int ih = q % g.Hi, it = q / g.Hi;
      // outputs o with o*s - p <= i < o*s - p + k
      const int ot0 = first_out(it + g.pt - g.kt, g.st);
      const int oh0 = first_out(ih + g.ph - g.kh, g.sh);
      const int ow0 = first_out(iw + g.pw - g.kw, g.sw);
      const int ot1 = min(g.To - 1, (it + g.pt) / g.st);
      const int oh1 = min(g.Ho - 1, (ih + g.ph) / g.sh);
      const int ow1 = min(g.Wo - 1, (iw + g.pw) / g.sw);
      float s = 0.f;
      for (int ot = ot0; ot <= ot1; ++ot)
        for (int oh = oh0; oh <= oh1; ++oh)
          for (int ow = ow0; ow <= ow1; ++ow) {
            const int o = (ot * g.Ho + oh) * g.Wo + ow;
            if (ip[o] == i) s += dyp[o];
          }
      dxp[i] = accumulate ? dxp[i] + s : s;
    }
  }
}

// y[n][c] = mean over S contiguous elements; one wave per (n, c).
__global__ void __launch_bounds__(256)
global_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int planes, int S) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= planes) return;
  const float* xp = x + (long)wave * S;
  float s = 0.f;
  for (int i = lane; i < S; i += 64) s += xp[i];
  s = wave_sum(s);
  if (lane == 0) y[wave] = s / (float)S;
}

__global__ void __launch_bounds__(256)
global_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long total, int S) {
  const float inv = 1.f / (float)S;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256)
    dx[e] = dy[e / S] * inv;
}

inline dim3 pool_grid(int planes, int elems) {
  int gx = cdiv(elems, 256 * 2);
  if (gx < 1) gx = 1;
  if (gx > 128) gx = 128;
  int gy = planes > 65535 ? 65535 : planes;
  while ((long)gx * gy > 262144 && gx > 1) gx >>= 1;
  return dim3(gx, gy);
}

inline PoolGeom to_geom(const coclr_pool_desc* d) {
  PoolGeom g;
  g.N = d->N; g.C = d->C;
  g.Ti = d->Ti; g.Hi = d->Hi; g.Wi = d->Wi; g.To = d->To; g.Ho = d->Ho; g.Wo = d->Wo;
  g.kt = d->kt; g.kh = d->kh; g.kw = d->kw; g.st = d->st; g.sh = d->sh; g.sw = d->sw;
  g.pt = d->pt; g.ph = d->ph; g.pw = d->pw;
  g.x_nstride = d->x_nstride; g.y_nstride = d->y_nstride;
  return g;
}

}  // namespace

extern "C" int coclr_maxpool3d_fwd(const coclr_pool_desc* d, const float* x, float* y,
                                   int32_t* indices, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  if (d->pt * 2 > d->kt || d->ph * 2 > d->kh || d->pw * 2 > d->kw) return COCLR_EINVAL;
  const PoolGeom g = to_geom(d);
  hipLaunchKernelGGL(maxpool3d_fwd_kernel, pool_grid(g.N * g.C, g.To * g.Ho * g.Wo), dim3(256), 0,
                     (hipStream_t)stream, x, y, indices, g);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_maxpool3d_bwd(const coclr_pool_desc* d, const float* dy, const int32_t* indices,
                                   float* dx, int64_t dy_nstride, int64_t dx_nstride,
                                   int accumulate, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  const PoolGeom g = to_geom(d);
  hipLaunchKernelGGL(maxpool3d_bwd_kernel, pool_grid(g.N * g.C, g.Ti * g.Hi * g.Wi), dim3(256), 0,
                     (hipStream_t)stream, dy, indices, dx, g, (long)dy_nstride, (long)dx_nstride,
                     accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_global_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t S,
                                        void* stream) {
  if (planes <= 0 || S <= 0) return COCLR_EINVAL;
  hipLaunchKernelGGL(global_avgpool_fwd_kernel, dim3(cdiv(planes * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, (int)planes, (int)S);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_global_avgpool_bwd(const float* dy, float* dx, int64_t planes, int64_t S,
                                        void* stream) {
  if (planes <= 0 || S <= 0) return COCLR_EINVAL;
  const long total = planes * S;
  int blocks = cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy,
                     dx, total, (int)S);
  COCLR_LAUNCH_CHECK();
  return 0;
}
