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
#include <cstdlib>

namespace {

struct PoolGeom {
  int N, C;
  int Ti, Hi, Wi, To, Ho, Wo;
  int kt, kh, kw, st, sh, sw, pt, ph, pw;
  long x_nstride, y_nstride;   // floats between samples (channel slices allowed)
  // optional transform of the INPUT, applied as it is read: x' = x*in_scale[c] + in_shift[c], then
  // max(x', 0) when in_relu -- the BatchNorm+ReLU that precedes the pool (backbone/s3dg.py:60-64
  // then :151,162; the block outputs before :173,190), so the normalised tensor is never written
  const float* in_scale;
  const float* in_shift;
  int in_relu;
  int C0;                      // channels before T was folded into the plane count
  int tfold;
  int nt;                      // stream the big operand with non-temporal loads / stores (see pool_nt_bytes)
};

typedef float pool_f32x4 __attribute__((ext_vector_type(4)));

// Non-temporal 16-byte access: the pooling passes over the stem's 1 GB tensors would otherwise evict what the
// MFMA-bound kernels of the other streams keep re-reading from L2 / the infinity cache (csrc/bn.hip, same switch)
__device__ __forceinline__ float4 pool_ld4(const float* p, int i, bool nt) {
  if (nt) {
    const pool_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const pool_f32x4*>(p) + i);
    return make_float4(t.x, t.y, t.z, t.w);
  }
  return reinterpret_cast<const float4*>(p)[i];
}
__device__ __forceinline__ void pool_st4(float* p, int i, const float4& w, bool nt) {
  if (nt) {
    pool_f32x4 t; t.x = w.x; t.y = w.y; t.z = w.z; t.w = w.w;
    __builtin_nontemporal_store(t, reinterpret_cast<pool_f32x4*>(p) + i);
  } else {
    reinterpret_cast<float4*>(p)[i] = w;
  }
}
inline long pool_nt_bytes() {
  const char* e = getenv("COCLR_BN_NT_MB");
  const long mb = e ? atol(e) : 0;
  return mb < 0 ? -1 : (mb << 20);
}

__device__ __forceinline__ float pool_in(const PoolGeom& g, float v, int c) {
  if (g.in_scale) {
    v = fmaf(v, g.in_scale[c], g.in_shift[c]);
    if (g.in_relu) v = fmaxf(v, 0.f);
  }
  return v;
}

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
            const float v = pool_in(g, xp[ii], c);
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
    const int ch = c / g.tfold;          // channel of the un-folded tensor
    if ((Si & 3) == 0 && ((g.x_nstride & 3) == 0)) {
      for (int i = threadIdx.x; i < (Si >> 2); i += 256) {
        float4 v = pool_ld4(xp, i, g.nt != 0);
        v.x = pool_in(g, v.x, ch); v.y = pool_in(g, v.y, ch);
        v.z = pool_in(g, v.z, ch); v.w = pool_in(g, v.w, ch);
        reinterpret_cast<float4*>(tp)[i] = v;
      }
    } else {
      for (int i = threadIdx.x; i < Si; i += 256) tp[i] = pool_in(g, xp[i], ch);
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

// Backward: the G input-gradient volumes live in LDS; every output element adds its gradient to
// the input element it selected.  Run-to-run DETERMINISTIC and atomic-free: outputs are visited in
// ceil(k/s)^3 colour classes ((ot, oh, ow) mod ceil(k/s)); the windows of two outputs of one class
// are disjoint, so within a class the read-modify-writes cannot collide, and the classes run in a
// fixed order with a barrier in between -- every input element receives its <= 27 addends in the
// same order every time.  Then the tile is written (or accumulated) to HBM with coalesced stores.
// With the class counts known at compile time (PT, PH, PW > 0) and at most KQ outputs per thread and
// class, every (arg-max, dy) pair of the workgroup is loaded UP FRONT -- one global round trip instead
// of one per class (27 for the 3x3x3 / 1 pools) -- and the classes are then applied from registers
// with an LDS-only barrier in between.  Same addends in the same order as the generic form.
template <int PT, int PH, int PW, int KQ>
struct ClassLoads {
  int i[PT * PH * PW][KQ];      // the arg-max index AS LOADED (-1: no element), added to b only when applied:
  int b[PT * PH * PW][KQ];      // arithmetic on the loaded value next to the load would make every class wait
  float v[PT * PH * PW][KQ];    // for its own round trip before the next class's loads are even issued
};

template <int PT, int PH, int PW, int KQ>
__device__ __forceinline__ void class_load(ClassLoads<PT, PH, PW, KQ>& L, const float* __restrict__ dy,
                                           const int* __restrict__ idx, const PoolGeom& g,
                                           long dy_nstride, int pl0, int gcount, int tfold, int Si,
                                           int So) {
#pragma unroll
  for (int ct = 0; ct < PT; ++ct)
#pragma unroll
    for (int ch = 0; ch < PH; ++ch)
#pragma unroll
      for (int cw = 0; cw < PW; ++cw) {
        const int cls = (ct * PH + ch) * PW + cw;
        const int nt = (g.To - ct + PT - 1) / PT, nh = (g.Ho - ch + PH - 1) / PH,
                  nw = (g.Wo - cw + PW - 1) / PW;
        const int csize = (nt > 0 && nh > 0 && nw > 0) ? nt * nh * nw : 0;
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
          const int q = threadIdx.x + k * 256;
          L.i[cls][k] = -1;
          L.b[cls][k] = 0;
          L.v[cls][k] = 0.f;
          if (q < gcount * csize) {
            const int gi = q / csize;
            int m = q - gi * csize;
            const int a = m / (nh * nw);
            m -= a * nh * nw;
            const int b = m / nw, cc = m - b * nw;
            const int o = ((a * PT + ct) * g.Ho + (b * PH + ch)) * g.Wo + (cc * PW + cw);
            const int pl = pl0 + gi;
            const int n = pl / g.C, c = pl - n * g.C;
            L.b[cls][k] = gi * Si - (pl % tfold) * Si;
            L.i[cls][k] = idx[(long)pl * So + o];
            L.v[cls][k] = dy[(long)n * dy_nstride + (long)c * So + o];
          }
        }
      }
}

// the tile's zero fill (or anything else written to LDS before) must be followed by no barrier of
// its own: the first class starts with one
template <int PT, int PH, int PW, int KQ>
__device__ __forceinline__ void class_apply(const ClassLoads<PT, PH, PW, KQ>& L, float* tile) {
#pragma unroll
  for (int cls = 0; cls < PT * PH * PW; ++cls) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the loads stay in flight
#pragma unroll
    for (int k = 0; k < KQ; ++k)
      if (L.i[cls][k] >= 0) tile[L.b[cls][k] + L.i[cls][k]] += L.v[cls][k];
  }
}

// generic form: class counts from the geometry, one global round trip per class
__device__ __forceinline__ void class_scatter_generic(float* tile, const float* __restrict__ dy,
                                                      const int* __restrict__ idx, const PoolGeom& g,
                                                      long dy_nstride, int pl0, int gcount, int tfold,
                                                      int Si, int So) {
  const int Pt = (g.kt + g.st - 1) / g.st, Ph = (g.kh + g.sh - 1) / g.sh, Pw = (g.kw + g.sw - 1) / g.sw;
  for (int ct = 0; ct < Pt; ++ct)
    for (int ch = 0; ch < Ph; ++ch)
      for (int cw = 0; cw < Pw; ++cw) {
        __syncthreads();
        const int nt = (g.To - ct + Pt - 1) / Pt, nh = (g.Ho - ch + Ph - 1) / Ph,
                  nw = (g.Wo - cw + Pw - 1) / Pw;
        if (nt <= 0 || nh <= 0 || nw <= 0) continue;       // uniform across the workgroup
        const int csize = nt * nh * nw;
        for (int q = threadIdx.x; q < gcount * csize; q += 256) {
          const int gi = q / csize;
          int m = q - gi * csize;
          const int a = m / (nh * nw);
          m -= a * nh * nw;
          const int b = m / nw, cc = m - b * nw;
          const int o = ((a * Pt + ct) * g.Ho + (b * Ph + ch)) * g.Wo + (cc * Pw + cw);
          const int pl = pl0 + gi;
          const int n = pl / g.C, c = pl - n * g.C;
          const int i = idx[(long)pl * So + o] - (pl % tfold) * Si;
          tile[gi * Si + i] += dy[(long)n * dy_nstride + (long)c * So + o];
        }
      }
}

template <int PT, int PH, int PW, int KQ>
__global__ void __launch_bounds__(256)
maxpool3d_tiled_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* dx,
                           const PoolGeom g, long dy_nstride, long dx_nstride, int accumulate,
                           int G, int planes, int tfold) {
  extern __shared__ float tile[];   // [G][Si]
  const int Si = g.Ti * g.Hi * g.Wi, So = g.To * g.Ho * g.Wo;
  const int pl0 = blockIdx.x * G;
  const int gcount = min(G, planes - pl0);
  if constexpr (PT > 0) {
    ClassLoads<PT, PH, PW, KQ> L;
    class_load<PT, PH, PW, KQ>(L, dy, idx, g, dy_nstride, pl0, gcount, tfold, Si, So);
    for (int i = threadIdx.x; i < gcount * Si; i += 256) tile[i] = 0.f;
    class_apply<PT, PH, PW, KQ>(L, tile);
  } else {
    for (int i = threadIdx.x; i < gcount * Si; i += 256) tile[i] = 0.f;
    class_scatter_generic(tile, dy, idx, g, dy_nstride, pl0, gcount, tfold, Si, So);
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


// ---------------------------------------------------------------------------
// 3x3x3 / stride 1 / pad 1 (the pool branch of every inception block, backbone/s3dg.py:105) on
// small planes (H*W = 256, 64 or 16).  The 27-element scan with argmax tracking is VALU-bound
// (~110 operations per output); the maximum is separable, and so is "first maximum in (t, h, w)
// scan order": max over w, then over h, then over t, each stage keeping the earliest of equal
// values -- 6 combines per output instead of 26.  Thread = (volume, h, w); it walks t with a
// three-plane sliding window of in-plane maxima.  NaN handling and the initial argmax (first
// in-range element) are those of the scan above.
// ---------------------------------------------------------------------------
struct PoolBest { float v; int i; };

__device__ __forceinline__ void pool_take(PoolBest& b, float cv, int ci) {
  if (cv > b.v || cv != cv) { b.v = cv; b.i = ci; }
}

template <bool WITH_IDX, int TT>
__global__ void __launch_bounds__(256)
maxpool333_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ idx,
                  const PoolGeom g, int planes, int lW, int lHW) {
  extern __shared__ float lds[];
  const int T = TT > 0 ? TT : g.Ti;                  // compile-time frame count: the t loops unroll
  const int W = 1 << lW, HW = 1 << lHW, H = HW >> lW;
  const int PG = 256 >> lHW, S = T << lHW;          // volumes per workgroup, floats per volume
  float* xs = lds;                                   // [PG][T][HW]
  float* bv = xs + PG * S;                           // w-stage maxima
  int* bi = reinterpret_cast<int*>(bv + PG * S);     // ... and their flat indices
  const int tid = threadIdx.x;
  const int pl0 = blockIdx.x * PG;
  const int gcount = min(PG, planes - pl0);
  if ((g.x_nstride & 3) == 0) {
    // [PG][S] is one run of <= 1024 float4 slots: all of a thread's loads in flight at once
    const int S4 = S >> 2, total4 = gcount * S4;
    for (int e0 = tid; e0 < total4; e0 += 4 * 256) {
      float4 r[4];
      int chn[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = e0 + k * 256;
        chn[k] = -1;
        if (e < total4) {
          const int gi = e / S4, i = e - gi * S4;
          const int pl = pl0 + gi;
          const int n = pl / g.C, c = pl - n * g.C;
          r[k] = reinterpret_cast<const float4*>(x + (long)n * g.x_nstride + (long)c * S)[i];
          chn[k] = c;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (chn[k] < 0) continue;
        float4 v = r[k];
        v.x = pool_in(g, v.x, chn[k]); v.y = pool_in(g, v.y, chn[k]);
        v.z = pool_in(g, v.z, chn[k]); v.w = pool_in(g, v.w, chn[k]);
        reinterpret_cast<float4*>(xs)[e0 + k * 256] = v;
      }
    }
  } else {
    for (int gi = 0; gi < gcount; ++gi) {
      const int pl = pl0 + gi;
      const int n = pl / g.C, c = pl - n * g.C;
      const float* xp = x + (long)n * g.x_nstride + (long)c * S;
      float* tp = xs + gi * S;
      for (int i = tid; i < S; i += 256) tp[i] = pool_in(g, xp[i], c);
    }
  }
  __syncthreads();
  const int gl = tid >> lHW, p = tid & (HW - 1), h = p >> lW, w = p & (W - 1);
  const bool live = gl < gcount;
  // stage 1: along w
  if (live) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = (gl * T + t) * HW + h * W;
      const int frow = (t * H + h) * W;              // flat index of the row start in the volume
      PoolBest b;
      b.v = -INFINITY;
      b.i = frow + (w > 0 ? w - 1 : 0);
      if (w > 0) pool_take(b, xs[row + w - 1], frow + w - 1);
      pool_take(b, xs[row + w], frow + w);
      if (w < W - 1) pool_take(b, xs[row + w + 1], frow + w + 1);
      bv[row + w] = b.v;
      if (WITH_IDX) bi[row + w] = b.i;
    }
  }
  __syncthreads();
  if (!live) return;
  // stage 2 (along h) for one plane
  auto plane_best = [&](int t) {
    const int col = (gl * T + t) * HW + w;
    const int h0 = h > 0 ? h - 1 : 0;
    PoolBest b;
    b.v = -INFINITY;
    b.i = WITH_IDX ? bi[col + h0 * W] : 0;
    if (h > 0) pool_take(b, bv[col + (h - 1) * W], WITH_IDX ? bi[col + (h - 1) * W] : 0);
    pool_take(b, bv[col + h * W], WITH_IDX ? bi[col + h * W] : 0);
    if (h < H - 1) pool_take(b, bv[col + (h + 1) * W], WITH_IDX ? bi[col + (h + 1) * W] : 0);
    return b;
  };
  const int pl = pl0 + gl;
  const int n = pl / g.C, c = pl - n * g.C;
  float* yp = y + (long)n * g.y_nstride + (long)c * S;
  int* ip = WITH_IDX ? idx + (long)pl * S : nullptr;
  // stage 3: sliding window along t
  PoolBest prev, cur = plane_best(0), next = cur;
  prev.v = -INFINITY; prev.i = cur.i;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) next = plane_best(t + 1);
    PoolBest b;
    b.v = -INFINITY;
    b.i = t > 0 ? prev.i : cur.i;                  // first in-range element of the window
    if (t > 0) pool_take(b, prev.v, prev.i);
    pool_take(b, cur.v, cur.i);
    if (t + 1 < T) pool_take(b, next.v, next.i);
    yp[t * HW + p] = b.v;
    if (WITH_IDX) ip[t * HW + p] = b.i;
    prev = cur;
    cur = next;
  }
}

// ---------------------------------------------------------------------------
// BatchNorm(+ReLU) backward of a unit whose ONLY consumer is a max-pool that applied the affine +
// ReLU while reading (engine.max_pool, lazy apply: MaxPool_2a behind Conv_1a, MaxPool_3a behind
// Conv_2c).  The gradient dz of the normalised activation is then nothing but the pool's dy scattered
// to the arg-max positions, so it is never materialised:
//   reduce (output-centric): sum g and sum g*xhat over the pool OUTPUTS, g = dy_pool[o] masked by the
//           ReLU at y[argmax(o)]  -- reads the two quarter-size tensors and gathers y;
//   apply  (the pool-backward tile kernel with a BatchNorm epilogue): the plane's g in LDS (colour
//           classes, deterministic), then dy_bn = A*g*mask + B*y + D streamed out with y streamed in.
// Against pool backward + two-pass BatchNorm backward (6.5 |y| of traffic, three kernels) this moves
// about 4 |y| in two.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bn_pool_bwd_reduce_kernel(const float* __restrict__ pdy, const int* __restrict__ pidx,
                          const float* __restrict__ y, const float* __restrict__ scale,
                          const float* __restrict__ shift, const float* __restrict__ mean,
                          const float* __restrict__ invstd, double* sums, int C, int So, int Si,
                          long pdy_nstride, long y_nstride, int relu) {
  __shared__ double red[4];
  const int c = blockIdx.x, n = blockIdx.y;
  const float sc = scale[c], sf = shift[c], mu = mean[c], is = invstd[c];
  const float* dp = pdy + (long)n * pdy_nstride + (long)c * So;
  const int* ip = pidx + ((long)n * C + c) * So;
  const float* yp = y + (long)n * y_nstride + (long)c * Si;
  float ag = 0.f, agx = 0.f;
  // four outputs per trip: the arg-max loads of all four, then the four dependent gathers of y -- two
  // round trips per four outputs instead of per output (same per-thread summation order)
  for (int o0 = threadIdx.x; o0 < So; o0 += 1024) {
    int ii[4];
    float gg[4], vv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int o = o0 + k * 256;
      ii[k] = o < So ? ip[o] : 0;
      gg[k] = o < So ? dp[o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) vv[k] = yp[ii[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (o0 + k * 256 >= So) break;
      float g = gg[k];
      const float v = vv[k];
      if (relu) g = fmaf(v, sc, sf) > 0.f ? g : 0.f;
      ag += g;
      agx += g * ((v - mu) * is);
    }
  }
  double sg = block256_sum_d((double)ag, red);
  double sgx = block256_sum_d((double)agx, red);
  if (threadIdx.x == 0) {
    sums[((long)c * gridDim.y + n) * 2] = sg;
    sums[((long)c * gridDim.y + n) * 2 + 1] = sgx;
  }
}

template <int PT, int PH, int PW, int KQ>
__global__ void __launch_bounds__(256)
bn_pool_bwd_apply_kernel(const float* __restrict__ pdy, const int* __restrict__ pidx,
                         const float* __restrict__ y, const float* __restrict__ scale,
                         const float* __restrict__ shift, const float* __restrict__ mean,
                         const float* __restrict__ invstd, const double* __restrict__ sums, int groups,
                         double count, int training, float* dgamma, float* dbeta, float* dy,
                         const PoolGeom g, long pdy_nstride, long y_nstride, long dy_nstride, int relu,
                         int G, int planes, int tfold) {
  extern __shared__ float tile[];   // [G][Si]
  __shared__ double tot[2];
  __shared__ float coef[8][4];      // per volume of the group: A, B, D, (unused)
  const int Si = g.Ti * g.Hi * g.Wi, So = g.To * g.Ho * g.Wo;
  const int pl0 = blockIdx.x * G;
  const int gcount = min(G, planes - pl0);
  ClassLoads<(PT > 0 ? PT : 1), (PT > 0 ? PH : 1), (PT > 0 ? PW : 1), (PT > 0 ? KQ : 1)> L;
  if constexpr (PT > 0)           // in flight while the coefficients are formed
    class_load<PT, PH, PW, KQ>(L, pdy, pidx, g, pdy_nstride, pl0, gcount, tfold, Si, So);
  for (int i = threadIdx.x; i < gcount * Si; i += 256) tile[i] = 0.f;
  // coefficients of the channels this group touches (g.C counts folded planes: channel = (pl % C) /
  // tfold); neighbouring planes of a folded volume share their channel's
  int cprev = -1;
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int c = (pl % g.C) / tfold;
    if (c == cprev && !(pl / g.C == 0 && (pl % g.C) % tfold == 0)) {      // uniform
      if (threadIdx.x == 0)       // the thread that wrote them
        for (int j = 0; j < 3; ++j) coef[gi][j] = coef[gi - 1][j];
      continue;
    }
    cprev = c;
    __syncthreads();
    if (threadIdx.x < 64) {
      double a0 = 0.0, a1 = 0.0;
      for (int k = threadIdx.x; k < groups; k += 64) {
        a0 += sums[((long)c * groups + k) * 2];
        a1 += sums[((long)c * groups + k) * 2 + 1];
      }
      a0 = wave_sum_d(a0);
      a1 = wave_sum_d(a1);
      if (threadIdx.x == 0) { tot[0] = a0; tot[1] = a1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double sg = tot[0], sgx = tot[1];
      const float sc = scale[c];
      float A = sc, B = 0.f, D = 0.f;
      if (training) {
        const float mg = (float)(sg / count), mgx = (float)(sgx / count);
        B = -sc * invstd[c] * mgx;
        D = sc * (mean[c] * invstd[c] * mgx - mg);
      }
      coef[gi][0] = A; coef[gi][1] = B; coef[gi][2] = D;
      // the plane (sample 0, frame 0) of a channel also emits its dgamma / dbeta
      if (pl / g.C == 0 && (pl % g.C) % tfold == 0) {
        if (dgamma) dgamma[c] = (float)sgx;
        if (dbeta) dbeta[c] = (float)sg;
      }
    }
  }
  if constexpr (PT > 0) class_apply<PT, PH, PW, KQ>(L, tile);
  else class_scatter_generic(tile, pdy, pidx, g, pdy_nstride, pl0, gcount, tfold, Si, So);
  __syncthreads();
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;       // c: folded plane index inside the sample
    const int ch = c / tfold;
    const float A = coef[gi][0], B = coef[gi][1], D = coef[gi][2];
    const float sc = scale[ch], sf = shift[ch];
    const float* yp = y + (long)n * y_nstride + (long)c * Si;
    float* dyp = dy + (long)n * dy_nstride + (long)c * Si;
    const float* tp = tile + gi * Si;
    if ((Si & 3) == 0 && ((y_nstride | dy_nstride) & 3) == 0) {
      for (int i = threadIdx.x; i < (Si >> 2); i += 256) {
        float4 gq = reinterpret_cast<const float4*>(tp)[i];
        const float4 v = pool_ld4(yp, i, g.nt != 0);
        if (relu) {
          gq.x = fmaf(v.x, sc, sf) > 0.f ? gq.x : 0.f; gq.y = fmaf(v.y, sc, sf) > 0.f ? gq.y : 0.f;
          gq.z = fmaf(v.z, sc, sf) > 0.f ? gq.z : 0.f; gq.w = fmaf(v.w, sc, sf) > 0.f ? gq.w : 0.f;
        }
        float4 o;
        o.x = fmaf(A, gq.x, fmaf(B, v.x, D)); o.y = fmaf(A, gq.y, fmaf(B, v.y, D));
        o.z = fmaf(A, gq.z, fmaf(B, v.z, D)); o.w = fmaf(A, gq.w, fmaf(B, v.w, D));
        pool_st4(dyp, i, o, g.nt != 0);
      }
    } else {
      for (int i = threadIdx.x; i < Si; i += 256) {
        float gv = tp[i];
        const float v = yp[i];
        if (relu) gv = fmaf(v, sc, sf) > 0.f ? gv : 0.f;
        dyp[i] = fmaf(A, gv, fmaf(B, v, D));
      }
    }
  }
}

// Backward of the 3x3x3 / stride 1 / pad 1 pool, GATHER form: dy and the argmax indices of G whole
// volumes are staged in LDS, thread i sums dy[o] over the <= 27 outputs o whose window holds input i
// and whose argmax is i, in fixed (t, h, w) order: no atomics, run-to-run deterministic, every
// thread busy (the scatter form has 27 colour classes of S/27 outputs each).
__global__ void __launch_bounds__(256)
maxpool333_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ idx, float* dx,
                      const PoolGeom g, long dy_nstride, long dx_nstride, int accumulate, int G,
                      int planes, int lW, int lHW) {
  extern __shared__ float lds[];
  const int T = g.Ti, W = 1 << lW, HW = 1 << lHW, H = HW >> lW;
  const int S = T << lHW;
  float* dys = lds;                                  // [G][S]
  int* ids = reinterpret_cast<int*>(lds + G * S);    // [G][S]
  const int pl0 = blockIdx.x * G;
  const int gcount = min(G, planes - pl0);
  for (int gi = 0; gi < gcount; ++gi) {
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    const float* dyp = dy + (long)n * dy_nstride + (long)c * S;
    const int* ip = idx + (long)pl * S;
    if ((dy_nstride & 3) == 0) {
      for (int i = threadIdx.x; i < (S >> 2); i += 256) {
        reinterpret_cast<float4*>(dys + gi * S)[i] = reinterpret_cast<const float4*>(dyp)[i];
        reinterpret_cast<int4*>(ids + gi * S)[i] = reinterpret_cast<const int4*>(ip)[i];
      }
    } else {
      for (int i = threadIdx.x; i < S; i += 256) { dys[gi * S + i] = dyp[i]; ids[gi * S + i] = ip[i]; }
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < gcount * S; q += 256) {
    const int gi = q / S, i = q - gi * S;
    const int t = i >> lHW, h = (i >> lW) & (H - 1), w = i & (W - 1);
    const float* dv = dys + gi * S;
    const int* iv = ids + gi * S;
    float s = 0.f;
#pragma unroll
    for (int dt = -1; dt <= 1; ++dt) {
      if (t + dt < 0 || t + dt >= T) continue;
#pragma unroll
      for (int dh = -1; dh <= 1; ++dh) {
        if (h + dh < 0 || h + dh >= H) continue;
#pragma unroll
        for (int dw = -1; dw <= 1; ++dw) {
          if (w + dw < 0 || w + dw >= W) continue;
          const int o = i + dt * HW + dh * W + dw;
          if (iv[o] == i) s += dv[o];
        }
      }
    }
    const int pl = pl0 + gi;
    const int n = pl / g.C, c = pl - n * g.C;
    float* d = dx + (long)n * dx_nstride + (long)c * S + i;
    *d = accumulate ? *d + s : s;
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
    static std::atomic<uint64_t> done{0};
    COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(k), kTileFloats * 4, done));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, stream, x, y, idx, g, G, planes, tfold);
  } else {
    auto k = maxpool3d_tiled_fwd_kernel<KT, KH, KW, ST, SH, SW, WPT, false>;
    static std::atomic<uint64_t> done{0};
    COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(k), kTileFloats * 4, done));
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
    g.tfold = g.Ti;
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
  g.in_scale = g.in_shift = nullptr; g.in_relu = 0; g.C0 = d->C; g.tfold = 1;
  {
    static const long nt_bytes = pool_nt_bytes();
    g.nt = (nt_bytes >= 0 && (long)d->N * d->C * d->Ti * d->Hi * d->Wi * 4 >= nt_bytes) ? 1 : 0;
  }
  return g;
}

}  // namespace

static inline int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

extern "C" int coclr_maxpool3d_fwd(const coclr_pool_desc* d, const float* x, float* y,
                                   int32_t* indices, const float* in_scale, const float* in_shift,
                                   int in_relu, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  if (d->pt * 2 > d->kt || d->ph * 2 > d->kh || d->pw * 2 > d->kw) return COCLR_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return COCLR_EINVAL;
  PoolGeom g = to_geom(d);
  g.in_scale = in_scale; g.in_shift = in_shift; g.in_relu = in_relu;
  if (g.kt == 3 && g.kh == 3 && g.kw == 3 && g.st == 1 && g.sh == 1 && g.sw == 1 && g.pt == 1 &&
      g.ph == 1 && g.pw == 1) {
    // inception pool branch: separable scan, thread = (volume, h, w)
    const int lW = ilog2_exact(g.Wi), lHW = ilog2_exact(g.Hi * g.Wi);
    if (lW >= 0 && lHW >= 4 && lHW <= 8 && g.Ti <= 32) {
      const int PG = 256 >> lHW, S = g.Ti << lHW;
      const int planes = g.N * g.C;
      const size_t lds = (size_t)PG * S * 4 * (indices ? 3 : 2);
      if (lds <= 64 * 1024) {
        const dim3 grid(cdiv(planes, PG));
        hipStream_t st = (hipStream_t)stream;
#define POOL333(TT)                                                                                  \
        if (indices) hipLaunchKernelGGL((maxpool333_kernel<true, TT>), grid, dim3(256), lds, st, x, y, \
                                        indices, g, planes, lW, lHW);                                \
        else hipLaunchKernelGGL((maxpool333_kernel<false, TT>), grid, dim3(256), lds, st, x, y,       \
                                indices, g, planes, lW, lHW)
        if (g.Ti == 16) { POOL333(16); }
        else if (g.Ti == 8) { POOL333(8); }
        else if (g.Ti == 4) { POOL333(4); }
        else { POOL333(0); }
#undef POOL333
        COCLR_LAUNCH_CHECK();
        return 0;
      }
    }
  }
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

// colour-class counts of a pool and the outputs a thread holds per class for G volumes per workgroup
struct ClassShape { int pt, ph, pw, kq; };
inline ClassShape class_shape(const PoolGeom& g, int G) {
  ClassShape c;
  c.pt = (g.kt + g.st - 1) / g.st; c.ph = (g.kh + g.sh - 1) / g.sh; c.pw = (g.kw + g.sw - 1) / g.sw;
  const long m = (long)((g.To + c.pt - 1) / c.pt) * ((g.Ho + c.ph - 1) / c.ph) * ((g.Wo + c.pw - 1) / c.pw);
  c.kq = (int)((G * m + 255) / 256);
  static const bool off = getenv("COCLR_POOL_PRELOAD") && atoi(getenv("COCLR_POOL_PRELOAD")) == 0;
  if (off) c.kq = 1 << 20;            // A/B switch: the generic one-round-trip-per-class form
  return c;
}

template <int PT, int PH, int PW, int KQ>
int launch_tiled_bwd(const float* dy, const int32_t* indices, float* dx, const PoolGeom& g,
                     long dy_nstride, long dx_nstride, int accumulate, int G, int planes, int tfold,
                     int Si, hipStream_t st) {
  auto kern = maxpool3d_tiled_bwd_kernel<PT, PH, PW, KQ>;
  static std::atomic<uint64_t> done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), kTileFloats * 4, done));
  hipLaunchKernelGGL(kern, dim3((planes + G - 1) / G), dim3(256), (size_t)G * Si * sizeof(float), st,
                     dy, indices, dx, g, dy_nstride, dx_nstride, accumulate, G, planes, tfold);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_maxpool3d_bwd(const coclr_pool_desc* d, const float* dy, const int32_t* indices,
                                   float* dx, int64_t dy_nstride, int64_t dx_nstride,
                                   int accumulate, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  const PoolGeom g0 = to_geom(d);
  if (g0.kt == 3 && g0.kh == 3 && g0.kw == 3 && g0.st == 1 && g0.sh == 1 && g0.sw == 1 &&
      g0.pt == 1 && g0.ph == 1 && g0.pw == 1) {
    const int lW = ilog2_exact(g0.Wi), lHW = ilog2_exact(g0.Hi * g0.Wi);
    const int S = g0.Ti * g0.Hi * g0.Wi;
    // measured (B=32): 4x4x4 volumes 0.027 ms gather vs 0.055 colour classes; 8x8x8 0.084 vs 0.075;
    // 16x16x16 0.327 vs 0.158 -- the 27-candidate scan is VALU-bound, it only pays on tiny volumes
    if (lW >= 0 && lHW >= 0 && S <= 128) {
      const int planes = g0.N * g0.C;
      int G = 2048 / S;              // ~16 KiB of LDS (dy + indices): 8 workgroups per CU
      if (G < 1) G = 1;
      while (G > 1 && (planes + G - 1) / G < 2048) G >>= 1;
      hipLaunchKernelGGL(maxpool333_bwd_kernel, dim3((planes + G - 1) / G), dim3(256),
                         (size_t)G * S * 8, (hipStream_t)stream, dy, indices, dx, g0,
                         (long)dy_nstride, (long)dx_nstride, accumulate, G, planes, lW, lHW);
      COCLR_LAUNCH_CHECK();
      return 0;
    }
  }
  {
    const PoolGeom g = fold_time(g0);
    const int tfold = g.Ti == g0.Ti ? 1 : g0.Ti;
    const int Si = g.Ti * g.Hi * g.Wi;
    const int planes = g.N * g.C;
    int G = pick_group(planes, Si);
    if (G > 0) {
      // a colour class holds ~1/27 of a volume's outputs: keep enough volumes per workgroup for its
      // 256 threads (1024 workgroups still fill the chip four deep)
      G = 4096 / Si > 0 ? 4096 / Si : 1;
      while (G > 1 && (planes + G - 1) / G < 1024) G >>= 1;
      const ClassShape cs = class_shape(g, G);
      hipStream_t st = (hipStream_t)stream;
#define POOL_BWD(a, b, c, q)                                                                         \
  if (cs.pt == a && cs.ph == b && cs.pw == c && cs.kq <= q)                                          \
    return launch_tiled_bwd<a, b, c, q>(dy, indices, dx, g, (long)dy_nstride, (long)dx_nstride,      \
                                        accumulate, G, planes, tfold, Si, st);
      POOL_BWD(1, 2, 2, 1) POOL_BWD(3, 3, 3, 1) POOL_BWD(2, 2, 2, 1) POOL_BWD(1, 1, 1, 2)
#undef POOL_BWD
      return launch_tiled_bwd<0, 0, 0, 0>(dy, indices, dx, g, (long)dy_nstride, (long)dx_nstride,
                                          accumulate, G, planes, tfold, Si, st);
    }
  }
  const PoolGeom& g = g0;
  hipLaunchKernelGGL(maxpool3d_bwd_kernel, pool_grid(g.N * g.C, g.Ti * g.Hi * g.Wi), dim3(256), 0,
                     (hipStream_t)stream, dy, indices, dx, g, (long)dy_nstride, (long)dx_nstride,
                     accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_bn_act_backward_pooled_fits(const coclr_pool_desc* d, int* fits) {
  if (!d || !fits || d->N <= 0 || d->C <= 0) return COCLR_EINVAL;
  const PoolGeom g = fold_time(to_geom(d));
  *fits = g.Ti * g.Hi * g.Wi <= kTileFloats;
  return 0;
}

extern "C" int coclr_bn_act_backward_pooled(const coclr_pool_desc* d, const float* pool_dy,
                                            const int32_t* pool_idx, const float* y,
                                            const float* scale, const float* shift,
                                            const float* mean, const float* invstd, double* sums,
                                            float* dy, float* dgamma, float* dbeta,
                                            int64_t pool_dy_nstride, int64_t y_nstride,
                                            int64_t dy_nstride, int relu, int training, void* stream) {
  if (!d || d->N <= 0 || d->C <= 0 || !pool_dy || !pool_idx || !y || !sums || !dy) return COCLR_EINVAL;
  const PoolGeom g0 = to_geom(d);
  const PoolGeom g = fold_time(g0);
  const int tfold = g.Ti == g0.Ti ? 1 : g0.Ti;
  const int Si = g.Ti * g.Hi * g.Wi;
  const int planes = g.N * g.C;
  if (Si > kTileFloats) return COCLR_EINVAL;          // see coclr_bn_act_backward_pooled_fits
  int G = 4096 / Si > 0 ? 4096 / Si : 1;
  if (G > 8) G = 8;
  while (G > 1 && (planes + G - 1) / G < 1024) G >>= 1;
  hipStream_t st = (hipStream_t)stream;
  const int So0 = g0.To * g0.Ho * g0.Wo, Si0 = g0.Ti * g0.Hi * g0.Wi;
  hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(g0.C, g0.N), dim3(256), 0, st, pool_dy, pool_idx,
                     y, scale, shift, mean, invstd, sums, g0.C, So0, Si0, (long)pool_dy_nstride,
                     (long)y_nstride, relu);
  COCLR_LAUNCH_CHECK();
  const double count = (double)g0.N * Si0;
  const ClassShape cs = class_shape(g, G);
#define BN_POOL_BWD(a, b, c, q)                                                                      \
  do {                                                                                               \
    auto kern = bn_pool_bwd_apply_kernel<a, b, c, q>;                                                \
    static std::atomic<uint64_t> done{0};                                                            \
    COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), kTileFloats * 4, done));      \
    hipLaunchKernelGGL(kern, dim3((planes + G - 1) / G), dim3(256), (size_t)G * Si * sizeof(float),  \
                       st, pool_dy, pool_idx, y, scale, shift, mean, invstd, sums, g0.N, count,      \
                       training, dgamma, dbeta, dy, g, (long)pool_dy_nstride, (long)y_nstride,       \
                       (long)dy_nstride, relu, G, planes, tfold);                                    \
    COCLR_LAUNCH_CHECK();                                                                            \
    return 0;                                                                                        \
  } while (0)
  if (cs.pt == 1 && cs.ph == 2 && cs.pw == 2 && cs.kq <= 1) BN_POOL_BWD(1, 2, 2, 1);
  if (cs.pt == 2 && cs.ph == 2 && cs.pw == 2 && cs.kq <= 1) BN_POOL_BWD(2, 2, 2, 1);
  BN_POOL_BWD(0, 0, 0, 0);
#undef BN_POOL_BWD
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
