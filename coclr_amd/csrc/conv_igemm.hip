// Implicit-GEMM 3D convolution for gfx950 on NCDHW fp32 tensors.
//
//   Y[n][co][o] = sum_{ci,tap} W[co][ci][tap] * V[n][ci][o*stride - pad + tap]
//
// where V is the input, optionally zero-upsampled ("input dilation") so the
// same kernel serves as the data-gradient of a strided convolution.  This one
// kernel family replaces every ATen/cuDNN conv3d the reference issues from
// backbone/s3dg.py:11-13,39-42 and backbone/resnet_2d3d.py:53-59,138 (forward),
// and the dgrad half of their autograd backward.
//
// Mapping to the hardware:
//   * GEMM view: M = Cout, N = output positions (a power-of-two 4-D "box"
//     n x t x h x w owned by one workgroup), K = Cin x taps.
//   * The input stencil window of the box is staged ONCE per Cin-chunk in LDS
//     ([c][window], positions contiguous); every tap reads it at a shifted
//     offset, so HBM/L2 sees each input element ~once per box instead of
//     once per tap.
//   * Weights arrive pre-packed as [tap][CinP][CoutP] (Cout contiguous) and are
//     staged as [tap][c][BM]; both MFMA operand reads are stride-1 across the
//     32 lanes of a half-wave -> conflict-free ds_read_b32.
//   * Math: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).  Lanes 0-31 carry
//     k = even channel of the chunk, lanes 32-63 the odd one, same tap.
//   * 256 threads = 4 waves as 2(M) x 2(N); next chunk is prefetched into
//     registers while the current one is multiplied.
//   * Epilogue: optional bias / per-channel affine / ReLU / accumulate, plus
//     per-workgroup partial sums (sum, sum of squares) per output channel for
//     train-mode BatchNorm, written without atomics as [2][Cout][ntiles].
#include "common.h"
#include "conv_geom.h"

namespace {

struct ConvArgs {
  const float* x;
  const float* w;        // packed [taps][CinP][CoutP]
  float* y;
  float* stats;          // [2][Cout][ntiles] or nullptr
  const float* bias;     // [Cout] or nullptr
  const float* ep_scale; // [Cout] or nullptr
  const float* ep_shift; // [Cout] or nullptr
  const int64_t* n_index;// optional gather of input samples
  long x_nstride, y_nstride;
  int x_cstride, y_cstride;
  int N, Cin, Cout, CinP, CoutP;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw, dt, dh, dw;
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int WT, WH, WW, plane1, plane;
  int mtiles, ntiles;
  int relu, accumulate;
};

template <int KT, int KH, int KW, int CC, int BM, int BN, int PT, int PI>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(const ConvArgs a) {
  constexpr int TAPS = KT * KH * KW;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BN / (WN * 32);
  constexpr int CG = 256 / PT;   // channel groups staged side by side
  constexpr int CI = CC / CG;    // channel iterations per thread per chunk
  static_assert(CC % CG == 0 && CC % 2 == 0, "chunk shape");
  static_assert(MF >= 1 && NF >= 1, "tile shape");
  constexpr int W4_TOTAL = TAPS * CC * BM / 4;
  constexpr int NW4 = (W4_TOTAL + 255) / 256;

  extern __shared__ __align__(16) float smem[];
  float* Ws = smem;                    // [TAPS*CC][BM]
  float* Xs = smem + TAPS * CC * BM;   // [CC][plane]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- which tile --------------------------------------------------------
  int bid = blockIdx.x;
  const int mt = bid % a.mtiles;
  const int ntile = bid / a.mtiles;
  int r = ntile;
  const int bw_ = r % a.nbw; r /= a.nbw;
  const int bh_ = r % a.nbh; r /= a.nbh;
  const int bt_ = r % a.nbt; r /= a.nbt;
  const int n0 = r << a.lTN;
  const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;
  const int cout0 = mt * BM;
  const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
  const int plane = a.plane;

  // ---- per-thread staging map for the input window ------------------------
  const int pe = tid % PT, cg = tid / PT;
  long goff[PI];
#pragma unroll
  for (int i = 0; i < PI; ++i) {
    const int e = pe + i * PT;
    goff[i] = -1;
    if (e < plane) {
      const int wn_ = e / a.plane1;
      int q = e - wn_ * a.plane1;
      const int hw = a.WH * a.WW;
      const int wt = q / hw; q -= wt * hw;
      const int wh = q / a.WW;
      const int ww = q - wh * a.WW;
      const int n = n0 + wn_;
      const int vt = vt0 + wt, vh = vh0 + wh, vw = vw0 + ww;
      if (n < a.N && vt >= 0 && vh >= 0 && vw >= 0) {
        const int it = vt / a.dt, ih = vh / a.dh, iw = vw / a.dw;
        if (it * a.dt == vt && ih * a.dh == vh && iw * a.dw == vw &&
            it < a.Ti && ih < a.Hi && iw < a.Wi) {
          const long ns = a.n_index ? (long)a.n_index[n] : (long)n;
          goff[i] = ns * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw;
        }
      }
    }
  }

  // ---- per-lane MFMA operand bases ----------------------------------------
  int lanebase[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BN / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = tn * a.plane1 + ((tt * a.st) * a.WH + th * a.sh) * a.WW + tw * a.sw +
                   half * plane;
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

  float xr[CI][PI];
  float4 wr[NW4];

  auto load_chunk = [&](int cin0) {
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const int cin = cin0 + ci * CG + cg;
      const bool cok = cin < a.Cin;
#pragma unroll
      for (int i = 0; i < PI; ++i) {
        float v = 0.f;
        if (cok && goff[i] >= 0) v = a.x[goff[i] + (long)cin * a.x_cstride];
        xr[ci][i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * 256;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e4 < W4_TOTAL) {
        const int row = e4 / (BM / 4), m4 = e4 % (BM / 4);
        const int tap = row / CC, c = row % CC;
        const int co = cout0 + m4 * 4;
        if (co < a.CoutP)
          v = *reinterpret_cast<const float4*>(
              a.w + ((long)tap * a.CinP + cin0 + c) * a.CoutP + co);
      }
      wr[i] = v;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) {
      const int c = ci * CG + cg;
#pragma unroll
      for (int i = 0; i < PI; ++i) {
        const int e = pe + i * PT;
        if (e < plane) Xs[c * plane + e] = xr[ci][i];
      }
    }
#pragma unroll
    for (int i = 0; i < NW4; ++i) {
      const int e4 = tid + i * 256;
      if (e4 < W4_TOTAL) *reinterpret_cast<float4*>(Ws + e4 * 4) = wr[i];
    }
  };

  const int nchunks = (a.Cin + CC - 1) / CC;
  load_chunk(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();   // previous chunk fully consumed
    store_chunk();
    __syncthreads();
    if (ch + 1 < nchunks) load_chunk((ch + 1) * CC);

#pragma unroll(KT > 3 && KH > 1 ? 1 : KT)
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
      for (int kh = 0; kh < KH; ++kh) {
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
          const int tap = (kt * KH + kh) * KW + kw;
          const int tapoff = (kt * a.WH + kh) * a.WW + kw;
#pragma unroll
          for (int q = 0; q < CC / 2; ++q) {
            float av[MF], bv[NF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
              av[mf] = Ws[abase + (tap * CC + 2 * q) * BM + mf * 32];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
              bv[nf] = Xs[lanebase[nf] + tapoff + 2 * q * plane];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
              for (int nf = 0; nf < NF; ++nf)
                acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    av[mf], bv[nf], acc[mf][nf], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------
  long yoff[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BN / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, ot = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    yoff[nf] = (n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo)
                   ? (long)n * a.y_nstride + ((long)ot * a.Ho + oh) * a.Wo + ow
                   : -1;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;  // [WN][BM][2], reused after the main loop
  if (want_stats) __syncthreads();

#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
      const int ml = wm * (BM / WM) + mf * 32 + row;
      const int co = cout0 + ml;
      const bool cok = co < a.Cout;
      float s = 0.f, ss = 0.f;
      float bia = 0.f, sc = 1.f, sf = 0.f;
      if (cok) {
        if (a.bias) bia = a.bias[co];
        if (a.ep_scale) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        if (cok && yoff[nf] >= 0) {
          float* dst = a.y + yoff[nf] + (long)co * a.y_cstride;
          float v = acc[mf][nf][i];
          if (a.accumulate) v += *dst;
          s += v; ss += v * v;
          v += bia;
          v = v * sc + sf;
          if (a.relu) v = fmaxf(v, 0.f);
          *dst = v;
        }
      }
      if (want_stats) {
        s = half_wave_sum(s);
        ss = half_wave_sum(ss);
        if (l31 == 0) {
          red[(wn * BM + ml) * 2 + 0] = s;
          red[(wn * BM + ml) * 2 + 1] = ss;
        }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        const float s = red[tid * 2] + red[(BM + tid) * 2];
        const float ss = red[tid * 2 + 1] + red[(BM + tid) * 2 + 1];
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}

// Weight re-layout:  dst[tap][r][c]  (r < RP rows = reduction channels,
// c < CP = produced channels), zero padded.
//   forward : r = cin,  c = cout, src tap = tap
//   dgrad   : r = cout, c = cin,  src tap = TAPS-1-tap (stencil flipped)
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ dst,
                                    int Cout, int Cin, int taps, long co_stride, long ci_stride,
                                    int tap_base, int RP, int CP, int transpose) {
  const long total = (long)taps * RP * CP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % CP);
    const long q = e / CP;
    const int rr = (int)(q % RP);
    const int tap = (int)(q / RP);
    float v = 0.f;
    if (!transpose) {
      if (rr < Cin && c < Cout) v = w[c * co_stride + rr * ci_stride + tap_base + tap];
    } else {
      if (rr < Cout && c < Cin) v = w[rr * co_stride + c * ci_stride + tap_base + (taps - 1 - tap)];
    }
    dst[e] = v;
  }
}

template <int KT, int KH, int KW, int CC, int BM, int BN, int PT, int PI>
int launch_variant(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  constexpr int TAPS = KT * KH * KW;
  if (p.plane > PT * PI) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, BM);
  const size_t lds_main = ((size_t)TAPS * CC * BM + (size_t)CC * p.plane) * sizeof(float);
  const size_t lds_red = (size_t)2 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  auto kern = conv_igemm_kernel<KT, KH, KW, CC, BM, BN, PT, PI>;
  static bool attr_done = false;
  if (!attr_done) {
    COCLR_RETURN_IF(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  const long blocks = (long)a.mtiles * a.ntiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// efficiency of covering Cout with tiles of BM rows
inline double cover(int cout, int bm) { return (double)cout / ((double)cdiv(cout, bm) * bm); }

struct Choice { int bm, lbn; };

// Decide (BM, BN) for a stencil class given which variants exist.
Choice choose_tile(const ConvPlan& base, int kt, int kh, int kw, bool has128x128, bool has64x64,
                   int max_plane_128, int max_plane_64) {
  Choice c{64, 7};
  ConvPlan p = base;
  conv_pick_box(&p, 7, kt, kh, kw);
  const bool fits128 = p.plane <= max_plane_128;
  long blocks64 = (long)p.ntiles * cdiv(base.Cout, 64);
  if (has128x128 && fits128 && cover(base.Cout, 128) >= cover(base.Cout, 64) - 1e-9 &&
      (long)p.ntiles * cdiv(base.Cout, 128) >= 384)
    c.bm = 128;
  if (!fits128 || (has64x64 && blocks64 < 384)) {
    ConvPlan p6 = base;
    conv_pick_box(&p6, 6, kt, kh, kw);
    if (has64x64 && p6.plane <= max_plane_64) { c.bm = 64; c.lbn = 6; }
  }
  return c;
}

}  // namespace

extern "C" int coclr_conv_packed_size(int cin, int cout, int taps, int transpose, int64_t* elems) {
  const int r = transpose ? cout : cin, c = transpose ? cin : cout;
  const long RP = ((r + 31) / 32) * 32, CP = ((c + 31) / 32) * 32;
  *elems = (int64_t)taps * RP * CP;
  return 0;
}

extern "C" int coclr_conv_pack_weights(const float* w, float* packed, int cout, int cin, int taps,
                                       int64_t co_stride, int64_t ci_stride, int tap_base,
                                       int transpose, void* stream) {
  const int r = transpose ? cout : cin, c = transpose ? cin : cout;
  const int RP = ((r + 31) / 32) * 32, CP = ((c + 31) / 32) * 32;
  const long total = (long)taps * RP * CP;
  int blocks = cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     packed, cout, cin, taps, (long)co_stride, (long)ci_stride, tap_base, RP, CP,
                     transpose);
  COCLR_LAUNCH_CHECK();
  return 0;
}

namespace {

// Fill plan + pick variant.  Returns 0 and the variant id, or an error.
int plan_forward(const coclr_conv_desc* d, ConvPlan* p, int* variant) {
  if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return COCLR_EINVAL;
  conv_normalise(d, p);
  const int kt = d->kt, kh = d->kh, kw = d->kw;
  Choice c;
  if (kt == 1 && kh == 1 && kw == 1) {
    const bool plain = p->st == 1 && p->sh == 1 && p->sw == 1 && p->dt == 1 && p->dh == 1 && p->dw == 1;
    if (plain) {
      c = choose_tile(*p, 1, 1, 1, true, true, 128, 64);
      conv_pick_box(p, c.lbn, 1, 1, 1);
      *variant = c.lbn == 6 ? 2 : (c.bm == 128 ? 0 : 1);
    } else {
      conv_pick_box(p, 6, 1, 1, 1);
      if (p->plane > 256) return COCLR_EINVAL;
      *variant = 3;
    }
  } else if (kt == 1 && kh == 3 && kw == 3) {
    c = choose_tile(*p, 1, 3, 3, true, true, 256, 512);
    conv_pick_box(p, c.lbn, 1, 3, 3);
    if (c.lbn == 6) *variant = p->plane <= 256 ? 12 : 13;
    else *variant = c.bm == 128 ? 10 : 11;
  } else if (kt == 3 && kh == 1 && kw == 1) {
    c = choose_tile(*p, 3, 1, 1, true, true, 256, 256);
    conv_pick_box(p, c.lbn, 3, 1, 1);
    *variant = c.lbn == 6 ? 22 : (c.bm == 128 ? 20 : 21);
  } else if (kt == 1 && kh == 7 && kw == 7) {
    conv_pick_box(p, 7, 1, 7, 7);
    *variant = 30;
  } else if (kt == 7 && kh == 1 && kw == 1) {
    conv_pick_box(p, 7, 7, 1, 1);
    *variant = 40;
  } else {
    return COCLR_EINVAL;
  }
  return 0;
}

}  // namespace

extern "C" int coclr_conv3d_ntiles(const coclr_conv_desc* d, int* ntiles) {
  ConvPlan p;
  int v;
  int rc = plan_forward(d, &p, &v);
  if (rc) return rc;
  *ntiles = p.ntiles;
  return 0;
}

extern "C" int coclr_conv3d_fwd(const coclr_conv_desc* d, const float* x, const float* w_packed,
                                float* y, float* stats, const float* bias, const float* ep_scale,
                                const float* ep_shift, const int64_t* n_index, int relu,
                                int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvPlan p;
  int variant;
  int rc = plan_forward(d, &p, &variant);
  if (rc) return rc;
  ConvArgs a;
  a.x = x; a.w = w_packed; a.y = y; a.stats = stats; a.bias = bias;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.n_index = n_index;
  a.x_nstride = d->x_nstride; a.y_nstride = d->y_nstride;
  a.x_cstride = p.Ti * p.Hi * p.Wi; a.y_cstride = p.To * p.Ho * p.Wo;
  a.N = p.N; a.Cin = p.Cin; a.Cout = p.Cout;
  a.CinP = ((p.Cin + 31) / 32) * 32; a.CoutP = ((p.Cout + 31) / 32) * 32;
  a.Ti = p.Ti; a.Hi = p.Hi; a.Wi = p.Wi; a.To = p.To; a.Ho = p.Ho; a.Wo = p.Wo;
  a.st = p.st; a.sh = p.sh; a.sw = p.sw; a.pt = p.pt; a.ph = p.ph; a.pw = p.pw;
  a.dt = p.dt; a.dh = p.dh; a.dw = p.dw;
  a.lTW = p.lTW; a.lTH = p.lTH; a.lTT = p.lTT; a.lTN = p.lTN;
  a.nbw = p.nbw; a.nbh = p.nbh; a.nbt = p.nbt; a.nbn = p.nbn;
  a.WT = p.WT; a.WH = p.WH; a.WW = p.WW; a.plane1 = p.plane1; a.plane = p.plane;
  a.ntiles = p.ntiles; a.mtiles = 0;
  a.relu = relu; a.accumulate = accumulate;
  switch (variant) {
    case 0:  return launch_variant<1, 1, 1, 32, 128, 128, 128, 1>(a, p, stream);
    case 1:  return launch_variant<1, 1, 1, 32, 64, 128, 128, 1>(a, p, stream);
    case 2:  return launch_variant<1, 1, 1, 32, 64, 64, 64, 1>(a, p, stream);
    case 3:  return launch_variant<1, 1, 1, 16, 64, 64, 256, 1>(a, p, stream);
    case 10: return launch_variant<1, 3, 3, 8, 128, 128, 256, 1>(a, p, stream);
    case 11: return launch_variant<1, 3, 3, 8, 64, 128, 256, 1>(a, p, stream);
    case 12: return launch_variant<1, 3, 3, 8, 64, 64, 256, 1>(a, p, stream);
    case 13: return launch_variant<1, 3, 3, 8, 64, 64, 256, 2>(a, p, stream);
    case 20: return launch_variant<3, 1, 1, 8, 128, 128, 256, 1>(a, p, stream);
    case 21: return launch_variant<3, 1, 1, 8, 64, 128, 256, 1>(a, p, stream);
    case 22: return launch_variant<3, 1, 1, 8, 64, 64, 256, 1>(a, p, stream);
    case 30: return launch_variant<1, 7, 7, 4, 64, 128, 256, 5>(a, p, stream);
    case 40: return launch_variant<7, 1, 1, 8, 64, 128, 256, 2>(a, p, stream);
  }
  return COCLR_EINVAL;
}
