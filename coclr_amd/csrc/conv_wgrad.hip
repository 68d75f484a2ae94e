// Weight gradient of the 3D convolution as a split-K implicit GEMM (gfx950):
//
//   dW[co][j] = sum_{n, o} dY[n][co][o] * X[n][ci(j)][o*stride - pad + tap(j)],
//   j = ci*taps + tap  (exactly the memory order of the [Cout][Cin][kt][kh][kw]
//   parameter, so the GEMM output IS the parameter-gradient layout).
//
// Replaces the wgrad half of ATen's convolution_backward for the modules in
// backbone/s3dg.py:11-13,39-42 and backbone/resnet_2d3d.py:53-59,138.
//
// GEMM view: M = Cout, N = J = Cin*taps, K = all output positions of the batch.
// One workgroup owns a (64 x BJ) tile of dW and a strided subset of the
// position boxes (split-K); per box it stages dY[64][128] (+1 pad, conflict
// free column reads) and the X stencil window of the channels its j-range
// touches, then runs v_mfma_f32_32x32x2_f32 with k = position.  Partial tiles
// go to [split][Cout][J]; a second kernel folds the splits (no atomics, so the
// result is run-to-run deterministic).
#include "common.h"
#include "conv_geom.h"

namespace {

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;            // [S][Cout][J]
  const int64_t* n_index; // optional gather of x samples (unused by callers today)
  long x_nstride, dy_nstride;
  int x_cstride, dy_cstride;
  int N, Cin, Cout, J, taps, KH, KW;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw;
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int WT, WH, WW, plane1, plane, planeP;
  int ntiles, S, jtiles, mtiles;
};

// BM = 64 couts, BN = 128 positions per staged box, BJ = j extent of the tile,
// NCI = max channels staged, PT/PI as in the forward kernel.
template <int BJ, int NCI, int PT, int PI>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradArgs a) {
  constexpr int BM = 64, BN = 128;
  constexpr int WM = 2, WN = 2;
  constexpr int NF = BJ / (WN * 32);
  constexpr int CG = 256 / PT;
  constexpr int CI = (NCI + CG - 1) / CG;
  constexpr int DYI = BM * BN / 256;   // dY elements per thread per box
  constexpr int LDY = BN + 1;

  extern __shared__ __align__(16) float smem[];
  float* dYs = smem;                       // [BM][LDY]
  int* pwoff = reinterpret_cast<int*>(smem + BM * LDY);  // [BN]
  float* Xs = smem + BM * LDY + BN;        // [NCI][planeP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int split = blockIdx.x;
  const int jt = blockIdx.y, mt = blockIdx.z;
  const int j0 = jt * BJ, cout0 = mt * BM;
  const int cin_lo = j0 / a.taps;
  int cin_hi = (j0 + BJ - 1) / a.taps;      // inclusive
  if (cin_hi >= a.Cin) cin_hi = a.Cin - 1;
  const int nci = cin_hi - cin_lo + 1;
  const int plane = a.plane, planeP = a.planeP;

  // box-position -> window offset table
  if (tid < BN) {
    const int p = tid;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    pwoff[p] = tn * a.plane1 + ((tt * a.st) * a.WH + th * a.sh) * a.WW + tw * a.sw;
  }

  // per-lane B bases: offset of (ci, tap) of column j inside Xs
  int jbase[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int j = j0 + wn * (BJ / WN) + nf * 32 + l31;
    int v = 0;
    if (j < a.J) {
      const int ci = j / a.taps, tap = j - ci * a.taps;
      const int kt = tap / (a.KH * a.KW);
      const int r = tap - kt * a.KH * a.KW;
      const int kh = r / a.KW, kw = r - kh * a.KW;
      v = (ci - cin_lo) * planeP + (kt * a.WH + kh) * a.WW + kw;
    }
    jbase[nf] = v;
  }
  const int abase = (wm * 32 + l31) * LDY;

  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

  const int pe = tid % PT, cg = tid / PT;
  const int pdy = tid & (BN - 1), mdy = tid >> 7;  // dY staging: position, row parity

  float xr[CI][PI];
  float dyr[DYI];

  auto load_tile = [&](int tile) {
    int r = tile;
    const int bw_ = r % a.nbw; r /= a.nbw;
    const int bh_ = r % a.nbh; r /= a.nbh;
    const int bt_ = r % a.nbt; r /= a.nbt;
    const int n0 = r << a.lTN;
    const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;
    // dY
    {
      const int p = pdy;
      const int tw = p & ((1 << a.lTW) - 1);
      const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
      const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
      const int tn = p >> (a.lTW + a.lTH + a.lTT);
      const int n = n0 + tn, ot = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
      const bool ok = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
      const long off = (long)n * a.dy_nstride + ((long)ot * a.Ho + oh) * a.Wo + ow;
#pragma unroll
      for (int i = 0; i < DYI; ++i) {
        const int co = cout0 + mdy + 2 * i;
        dyr[i] = (ok && co < a.Cout) ? a.dy[off + (long)co * a.dy_cstride] : 0.f;
      }
    }
    // X window
    const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
#pragma unroll
    for (int i = 0; i < PI; ++i) {
      const int e = pe + i * PT;
      long g = -1;
      if (e < plane) {
        const int wn_ = e / a.plane1;
        int q = e - wn_ * a.plane1;
        const int hw = a.WH * a.WW;
        const int wt = q / hw; q -= wt * hw;
        const int wh = q / a.WW;
        const int ww = q - wh * a.WW;
        const int n = n0 + wn_;
        const int it = vt0 + wt, ih = vh0 + wh, iw = vw0 + ww;
        if (n < a.N && it >= 0 && ih >= 0 && iw >= 0 && it < a.Ti && ih < a.Hi && iw < a.Wi) {
          const long ns = a.n_index ? (long)a.n_index[n] : (long)n;
          g = ns * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw;
        }
      }
#pragma unroll
      for (int ci = 0; ci < CI; ++ci) {
        const int c = ci * CG + cg;
        xr[ci][i] = (g >= 0 && c < nci) ? a.x[g + (long)(cin_lo + c) * a.x_cstride] : 0.f;
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < DYI; ++i) dYs[(mdy + 2 * i) * LDY + pdy] = dyr[i];
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const int c = ci * CG + cg;
      if (c < NCI) {
#pragma unroll
        for (int i = 0; i < PI; ++i) {
          const int e = pe + i * PT;
          if (e < plane) Xs[c * planeP + e] = xr[ci][i];
        }
      }
    }
  };

  int tile = split;
  if (tile < a.ntiles) load_tile(tile);
  for (; tile < a.ntiles; tile += a.S) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + a.S < a.ntiles) load_tile(tile + a.S);
#pragma unroll 8
    for (int s = 0; s < BN / 2; ++s) {
      const int p = 2 * s + half;
      const float av = dYs[abase + p];
      const int wo = pwoff[p];
      float bv[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bv[nf] = Xs[jbase[nf] + wo];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[nf], acc[nf], 0, 0, 0);
    }
  }

  // partial tile out: part[split][co][j]
  float* out = a.part + (long)split * a.Cout * a.J;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
    const int co = cout0 + wm * 32 + row;
    if (co < a.Cout) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int j = j0 + wn * (BJ / WN) + nf * 32 + l31;
        if (j < a.J) out[(long)co * a.J + j] = acc[nf][i];
      }
    }
  }
}

// dW[co][ci*ci_stride' ...] = sum_s part[s][e]; destination may be a tap slice of a
// larger stencil (r50 stem): e = co*J + ci*taps + tap ->
// dst[co*co_stride + ci*ci_stride + tap_base + tap].
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long CJ,
                                    int S, int J, int taps, long co_stride, long ci_stride,
                                    int tap_base, int accumulate) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < CJ;
       e += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += part[(long)k * CJ + e];
    const long co = e / J;
    const int j = (int)(e - co * J);
    const int ci = j / taps, tap = j - ci * taps;
    float* d = dw + co * co_stride + ci * ci_stride + tap_base + tap;
    *d = accumulate ? *d + s : s;
  }
}

struct WPlan {
  ConvPlan p;
  int variant, BJ, S, jtiles, mtiles, planeP;
};

int plan_wgrad(const coclr_conv_desc* d, WPlan* w) {
  if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return COCLR_EINVAL;
  if (d->dt != 1 || d->dh != 1 || d->dw != 1) return COCLR_EINVAL;
  ConvPlan& p = w->p;
  conv_normalise(d, &p);
  const int taps = d->kt * d->kh * d->kw;
  conv_pick_box(&p, 7, d->kt, d->kh, d->kw);
  const int J = p.Cin * taps;
  // variant table: (BJ, NCI, PT, PI)
  if (taps == 1) {
    if (p.plane <= 128) { w->variant = 0; w->BJ = 64; }
    else if (p.plane <= 256) { w->variant = 1; w->BJ = 64; }
    else return COCLR_EINVAL;
  } else if (taps == 3) {
    if (p.plane > 256) return COCLR_EINVAL;
    w->variant = 2; w->BJ = 128;       // NCI = 44
  } else if (taps == 9) {
    if (p.plane <= 256) w->variant = 3;
    else if (p.plane <= 768) w->variant = 4;
    else if (p.plane <= 1024) w->variant = 7;
    else return COCLR_EINVAL;
    w->BJ = 128;                        // NCI = 16
  } else if (taps == 7) {
    if (p.plane > 512) return COCLR_EINVAL;
    w->variant = 5; w->BJ = 128;        // NCI = 20
  } else if (taps == 49) {
    if (p.plane > 1280) return COCLR_EINVAL;
    w->variant = 6; w->BJ = 128;        // NCI = 4
  } else {
    return COCLR_EINVAL;
  }
  w->planeP = p.plane | 1;
  w->jtiles = cdiv(J, w->BJ);
  w->mtiles = cdiv(p.Cout, 64);
  int S = 1536 / (w->jtiles * w->mtiles);
  if (S < 1) S = 1;
  if (S > p.ntiles) S = p.ntiles;
  w->S = S;
  return 0;
}

template <int BJ, int NCI, int PT, int PI>
int launch_wgrad(WgradArgs& a, hipStream_t stream) {
  const size_t lds = ((size_t)64 * 129 + 128 + (size_t)NCI * a.planeP) * sizeof(float);
  if (lds > 160 * 1024) return COCLR_EINVAL;
  auto kern = conv_wgrad_kernel<BJ, NCI, PT, PI>;
  static bool attr_done = false;
  if (!attr_done) {
    COCLR_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.S, a.jtiles, a.mtiles), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int coclr_conv3d_wgrad_workspace(const coclr_conv_desc* d, int64_t* elems) {
  WPlan w;
  int rc = plan_wgrad(d, &w);
  if (rc) return rc;
  *elems = (int64_t)w.S * d->Cout * d->Cin * d->kt * d->kh * d->kw;
  return 0;
}

extern "C" int coclr_conv3d_wgrad(const coclr_conv_desc* d, const float* x, const float* dy,
                                  float* dw, float* workspace, int64_t w_co_stride,
                                  int64_t w_ci_stride, int tap_base, int accumulate,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  WPlan w;
  int rc = plan_wgrad(d, &w);
  if (rc) return rc;
  const ConvPlan& p = w.p;
  WgradArgs a;
  a.x = x; a.dy = dy; a.part = workspace; a.n_index = nullptr;
  a.x_nstride = d->x_nstride; a.dy_nstride = d->y_nstride;
  a.x_cstride = p.Ti * p.Hi * p.Wi; a.dy_cstride = p.To * p.Ho * p.Wo;
  a.N = p.N; a.Cin = p.Cin; a.Cout = p.Cout;
  a.taps = d->kt * d->kh * d->kw; a.KH = d->kh; a.KW = d->kw;
  a.J = p.Cin * a.taps;
  a.Ti = p.Ti; a.Hi = p.Hi; a.Wi = p.Wi; a.To = p.To; a.Ho = p.Ho; a.Wo = p.Wo;
  a.st = p.st; a.sh = p.sh; a.sw = p.sw; a.pt = p.pt; a.ph = p.ph; a.pw = p.pw;
  a.lTW = p.lTW; a.lTH = p.lTH; a.lTT = p.lTT; a.lTN = p.lTN;
  a.nbw = p.nbw; a.nbh = p.nbh; a.nbt = p.nbt; a.nbn = p.nbn;
  a.WT = p.WT; a.WH = p.WH; a.WW = p.WW; a.plane1 = p.plane1; a.plane = p.plane;
  a.planeP = w.planeP;
  a.ntiles = p.ntiles; a.S = w.S; a.jtiles = w.jtiles; a.mtiles = w.mtiles;
  switch (w.variant) {
    case 0: rc = launch_wgrad<64, 64, 128, 1>(a, stream); break;
    case 1: rc = launch_wgrad<64, 64, 256, 1>(a, stream); break;
    case 2: rc = launch_wgrad<128, 44, 256, 1>(a, stream); break;
    case 3: rc = launch_wgrad<128, 16, 256, 1>(a, stream); break;
    case 4: rc = launch_wgrad<128, 16, 256, 3>(a, stream); break;
    case 5: rc = launch_wgrad<128, 20, 256, 2>(a, stream); break;
    case 6: rc = launch_wgrad<128, 4, 256, 5>(a, stream); break;
    case 7: rc = launch_wgrad<128, 16, 256, 4>(a, stream); break;
    default: rc = COCLR_EINVAL;
  }
  if (rc) return rc;
  const long CJ = (long)a.Cout * a.J;
  int blocks = cdiv(CJ, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, dw, CJ,
                     a.S, a.J, a.taps, (long)w_co_stride, (long)w_ci_stride, tap_base, accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}
