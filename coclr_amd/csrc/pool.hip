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

// ---------------------------------------------------------------------------
// LDS-tiled pooling for the configurations the backbones use.  One workgroup
// owns G whole (n, c) input volumes (<= 16384 floats together): the volume is
// read from HBM exactly once with coalesced loads, the window scan runs out of
// LDS with a compile-time stencil, each lane produces WPT adjacent outputs of
// one row so every LDS row segment is read once per WPT outputs.
// ---------------------------------------------------------------------------
template <int KT, int KH, int KW, int ST, int SH, int SW, int WPT, bool WITH_IDX>
__global__ void __launch_bounds__(256)
maxpool3d_tiled_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                           int* __restrict__ idx, const PoolGeom g, int G, int planes, int tfold) {
  extern __shared__ float tile[];   // [G][Si]
  const int Si = g.Ti * g.Hi * g.Wi, So = g.To * g.Ho * g.Wo;
  const int pl0 = blockIdx.x * G;
  const int gcount = min(G, planes - pl0);
  // ---- load: G volumes are contiguous per sample only if they share n; handle generally
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    const float* xp = x + (long)n * g.x_nstride + (long)c * Si;
    float* tp = tile + gi * Si;
    if ((Si & 3) == 0 && ((g.x_nstride & 3) == 0)) {
      for (int i = threadIdx.x; i < (Si >> 2); i += 256)
        reinterpret_cast<float4*>(tp)[i] = reinterpret_cast<const float4*>(xp)[i];
    } else {
      for (int i = threadIdx.x; i < Si; i += 256) tp[i] = xp[i];
    }
  }
  __syncthreads();
  constexpr int NR = (WPT - 1) * SW + KW;    // input columns feeding WPT outputs
  const int OWC = (g.Wo + WPT - 1) / WPT;
  const int rows = g.To * g.Ho;
  const int items = gcount * rows * OWC;
  for (int it = threadIdx.x; it < items; it += 256) {
    const int owc = it % OWC;
    int r = it / OWC;
    const int oh = r % g.Ho; r /= g.Ho;
    const int ot = r % g.To;
    const int gi = r / g.To;
    const int ow0 = owc * WPT;
    const float* tp = tile + gi * Si;
    float best[WPT];
    int bi[WPT];
    const int t0 = ot * ST - g.pt, h0 = oh * SH - g.ph, w0 = ow0 * SW - g.pw;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      best[j] = -INFINITY;
      // ATen initialises argmax to the first in-range window element
      const int tt = max(t0, 0), hh = max(h0, 0), ww = max(w0 + j * SW, 0);
      bi[j] = (tt * g.Hi + hh) * g.Wi + ww;
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int t = t0 + kt;
      if (t < 0 || t >= g.Ti) continue;
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
        const int h = h0 + kh;
        if (h < 0 || h >= g.Hi) continue;
        const int rowoff = (t * g.Hi + h) * g.Wi;
        float v[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
          const int w = w0 + q;
          v[q] = (w >= 0 && w < g.Wi) ? tp[rowoff + w] : -INFINITY;
        }
#pragma unroll
        for (int j = 0; j < WPT; ++j) {
#pragma unroll
          for (int kw = 0; kw < KW; ++kw) {
            const float val = v[j * SW + kw];
            if (val > best[j] || val != val) { best[j] = val; bi[j] = rowoff + w0 + j * SW + kw; }
          }
        }
      }
    }
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    const long obase = ((long)ot * g.Ho + oh) * g.Wo + ow0;
    float* yp = y + (long)n * g.y_nstride + (long)c * So + obase;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      if (ow0 + j < g.Wo) {
        yp[j] = best[j];
        // indices are relative to the un-folded (n, c) volume
        if (WITH_IDX) idx[(long)pl * So + obase + j] = bi[j] + (pl % tfold) * Si;
      }
    }
  }
}

// Backward: the G input-gradient volumes live in LDS; every output element scatters its
// gradient with one LDS float atomic (order of the <= 27 addends per element is not
// fixed: results are reproducible to fp32 round-off, not bitwise), then the tile is
// written (or accumulated) to HBM with coalesced stores.
__global__ void __launch_bounds__(256)
maxpool3d_tiled_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* dx,
                           const PoolGeom g, long dy_nstride, long dx_nstride, int accumulate,
                           int G, int planes, int tfold) {
  extern __shared__ float tile[];   // [G][Si]
  const int Si = g.Ti * g.Hi * g.Wi, So = g.To * g.Ho * g.Wo;
  const int pl0 = blockIdx.x * G;
  const int gcount = min(G, planes - pl0);
  for (int i = threadIdx.x; i < gcount * Si; i += 256) tile[i] = 0.f;
  __syncthreads();
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    const float* dyp = dy + (long)n * dy_nstride + (long)c * So;
    const int* ip = idx + (long)pl * So;
    float* tp = tile + gi * Si;
    const int rebase = (pl % tfold) * Si;
    for (int o = threadIdx.x; o < So; o += 256) atomicAdd(&tp[ip[o] - rebase], dyp[o]);
  }
  __syncthreads();
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    float* dxp = dx + (long)n * dx_nstride + (long)c * Si;
    const float* tp = tile + gi * Si;
    if ((Si & 3) == 0 && ((dx_nstride & 3) == 0)) {
      for (int i = threadIdx.x; i < (Si >> 2); i += 256) {
        float4 v = reinterpret_cast<const float4*>(tp)[i];
        if (accumulate) {
          const float4 o = reinterpret_cast<const float4*>(dxp)[i];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        reinterpret_cast<float4*>(dxp)[i] = v;
      }
    } else {
      for (int i = threadIdx.x; i < Si; i += 256) dxp[i] = accumulate ? dxp[i] + tp[i] : tp[i];
    }
  }
}

constexpr int kTileFloats = 16384;   // 64 KiB of LDS per workgroup

// planes per workgroup so that the tile is <= kTileFloats and there are enough workgroups
inline int pick_group(int planes, int Si) {
  if (Si > kTileFloats) return 0;
  // ~16 KiB tiles: 8 workgroups per CU so the load, scan and store phases of different
  // workgroups overlap (64 KiB tiles left 2 per CU and ran at 1.2 TB/s)
  int G = 4096 / Si;
  if (G < 1) G = 1;
  while (G > 1 && (planes + G - 1) / G < 2048) G >>= 1;
  return G;
}

template <int KT, int KH, int KW, int ST, int SH, int SW, int WPT>
int launch_tiled_fwd(const PoolGeom& g, const float* x, float* y, int* idx, int G, int planes,
                     int tfold, hipStream_t stream) {
  const size_t lds = (size_t)G * g.Ti * g.Hi * g.Wi * sizeof(float);
  const int blocks = (planes + G - 1) / G;
  if (idx) {
    auto k = maxpool3d_tiled_fwd_kernel<KT, KH, KW, ST, SH, SW, WPT, true>;
    static bool done = false;
    if (!done) { COCLR_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kTileFloats * 4)); done = true; }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, stream, x, y, idx, g, G, planes, tfold);
  } else {
    auto k = maxpool3d_tiled_fwd_kernel<KT, KH, KW, ST, SH, SW, WPT, false>;
    static bool done = false;
    if (!done) { COCLR_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kTileFloats * 4)); done = true; }
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, stream, x, y, idx, g, G, planes, tfold);
  }
  COCLR_LAUNCH_CHECK();
  return 0;
}

// Collapse T into the plane count when the stencil does not touch it.
inline PoolGeom fold_time(PoolGeom g) {
  if (g.kt == 1 && g.st == 1 && g.pt == 0 && g.Ti == g.To) {
    // (n, c, t) volumes of H*W: only valid when samples are dense in (C,T,H,W), which
    // holds for channel-slice views as long as C*T planes of one sample are contiguous
    g.C = g.C * g.Ti;
    g.Ti = g.To = 1;
  }
  return g;
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
  {
    // LDS-tiled fast paths; T is folded into the plane count when the stencil ignores it
    const PoolGeom f = fold_time(g);
    const int tfold = f.Ti == g.Ti ? 1 : g.Ti;
    const int Si = f.Ti * f.Hi * f.Wi;
    const int planes = f.N * f.C;
    const int G = pick_group(planes, Si);
    hipStream_t st = (hipStream_t)stream;
    if (G > 0) {
      const int k[3] = {f.kt, f.kh, f.kw}, s[3] = {f.st, f.sh, f.sw};
#define POOL_IS(a, b, c, e, ff, gg) (k[0] == a && k[1] == b && k[2] == c && s[0] == e && s[1] == ff && s[2] == gg)
      if (POOL_IS(1, 3, 3, 1, 2, 2)) return launch_tiled_fwd<1, 3, 3, 1, 2, 2, 2>(f, x, y, indices, G, planes, tfold, st);
      if (POOL_IS(3, 3, 3, 1, 1, 1)) return launch_tiled_fwd<3, 3, 3, 1, 1, 1, 4>(f, x, y, indices, G, planes, tfold, st);
      if (POOL_IS(3, 3, 3, 2, 2, 2)) return launch_tiled_fwd<3, 3, 3, 2, 2, 2, 2>(f, x, y, indices, G, planes, tfold, st);
      if (POOL_IS(2, 2, 2, 2, 2, 2)) return launch_tiled_fwd<2, 2, 2, 2, 2, 2, 2>(f, x, y, indices, G, planes, tfold, st);
#undef POOL_IS
    }
  }
  hipLaunchKernelGGL(maxpool3d_fwd_kernel, pool_grid(g.N * g.C, g.To * g.Ho * g.Wo), dim3(256), 0,
                     (hipStream_t)stream, x, y, indices, g);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_maxpool3d_bwd(const coclr_pool_desc* d, const float* dy, const int32_t* indices,
                                   float* dx, int64_t dy_nstride, int64_t dx_nstride,
                                   int accumulate, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  const PoolGeom g0 = to_geom(d);
  {
    const PoolGeom g = fold_time(g0);
    const int tfold = g.Ti == g0.Ti ? 1 : g0.Ti;
    const int Si = g.Ti * g.Hi * g.Wi;
    const int planes = g.N * g.C;
    const int G = pick_group(planes, Si);
    if (G > 0) {
      static bool done = false;
      if (!done) {
        COCLR_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void*>(maxpool3d_tiled_bwd_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kTileFloats * 4));
        done = true;
      }
      hipLaunchKernelGGL(maxpool3d_tiled_bwd_kernel, dim3((planes + G - 1) / G), dim3(256),
                         (size_t)G * Si * sizeof(float), (hipStream_t)stream, dy, indices, dx, g,
                         (long)dy_nstride, (long)dx_nstride, accumulate, G, planes, tfold);
      COCLR_LAUNCH_CHECK();
      return 0;
    }
  }
  const PoolGeom& g = g0;
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
