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
#include <cstdlib>

#include "common.h"
#include "conv_geom.h"

namespace {

__device__ float g_zero[64];   // zero-initialised; source for masked lanes of the LDS DMA

struct WgradArgs {
  const float* x;
  const float* dy;
  float* part;            // [S][Cout][J]
  long x_nstride, dy_nstride;
  int x_cstride, dy_cstride;
  int N, Cin, Cout, J, taps, KH, KW;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw;
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int WT, WH, WW, plane1, plane, planeP, pch;
  int ntiles, S, jtiles, mtiles;
};

#define GLDS4(gptr, lptr)                                                                   \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),   \
                                   (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// BM = 64 couts, BN = 128 positions per staged box, BJ = j extent of the tile,
// PCH = max number of 64-element chunks of one channel's window.
// Staging is done entirely by the LDS DMA engine (global_load_lds_dword): no staging
// registers, every load of a tile in flight at once; masked elements read g_zero.
template <int BJ, int PCH>
__global__ void __launch_bounds__(256)
conv_wgrad_kernel(const WgradArgs a) {
  constexpr int BM = 64, BN = 128;
  constexpr int WN = 2;
  constexpr int NF = BJ / (WN * 32);
  constexpr int LDY = BN + 1;

  extern __shared__ __align__(16) float smem[];
  float* dYs = smem;                                      // [BM][LDY]
  int* pwoff = reinterpret_cast<int*>(smem + BM * LDY);   // [BN]
  float* Xs = smem + BM * LDY + BN;                       // [nci][planeP]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  const int split = blockIdx.x;
  const int j0 = blockIdx.y * BJ, cout0 = blockIdx.z * BM;
  const int cin_lo = j0 / a.taps;
  int cin_hi = (j0 + BJ - 1) / a.taps;      // inclusive
  if (cin_hi >= a.Cin) cin_hi = a.Cin - 1;
  const int nci = cin_hi - cin_lo + 1;
  const int planeP = a.planeP;
  const int lW = a.lTW, lWH = a.lTW + a.lTH, lWHT = a.lTW + a.lTH + a.lTT;

  // box-position -> window offset table
  if (tid < BN) {
    const int p = tid;
    const int tw = p & ((1 << lW) - 1);
    const int th = (p >> lW) & ((1 << a.lTH) - 1);
    const int tt = (p >> lWH) & ((1 << a.lTT) - 1);
    const int tn = p >> lWHT;
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

  // tile-independent decode of this lane's window elements: packed (wn, wt, wh, ww)
  unsigned wcoord[PCH];
#pragma unroll
  for (int ch = 0; ch < PCH; ++ch) {
    const int e = ch * 64 + lane;
    unsigned v = 0xffffffffu;
    if (e < a.plane) {
      const int wn_ = e / a.plane1;
      int q = e - wn_ * a.plane1;
      const int hw = a.WH * a.WW;
      const int wt = q / hw; q -= wt * hw;
      const int wh = q / a.WW;
      const int ww = q - wh * a.WW;
      v = (unsigned)wn_ | ((unsigned)wt << 8) | ((unsigned)wh << 16) | ((unsigned)ww << 24);
    }
    wcoord[ch] = v;
  }

  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

  for (int tile = split; tile < a.ntiles; tile += a.S) {
    int r = tile;
    const int bw_ = r % a.nbw; r /= a.nbw;
    const int bh_ = r % a.nbh; r /= a.nbh;
    const int bt_ = r % a.nbt; r /= a.nbt;
    const int n0 = r << a.lTN;
    const int ow0 = bw_ << lW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;

    __syncthreads();     // previous tile fully consumed (also orders pwoff on the first pass)

    // ---- dY: rows wave, wave+4, ... ; two 64-position halves per row ----------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = h * 64 + lane;
      const int tw = p & ((1 << lW) - 1);
      const int th = (p >> lW) & ((1 << a.lTH) - 1);
      const int tt = (p >> lWH) & ((1 << a.lTT) - 1);
      const int tn = p >> lWHT;
      const int n = n0 + tn, ot = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
      const bool ok = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
      const float* base = a.dy + (long)n * a.dy_nstride + ((long)ot * a.Ho + oh) * a.Wo + ow;
      for (int row = wave; row < BM; row += 4) {
        const int co = cout0 + row;
        const float* src = (ok && co < a.Cout) ? base + (long)co * a.dy_cstride : g_zero;
        GLDS4(src, dYs + row * LDY + h * 64);
      }
    }
    // ---- X window: channels wave, wave+4, ... ----------------------------------------
    const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
#pragma unroll
    for (int ch = 0; ch < PCH; ++ch) {
      if (ch < a.pch) {
        const unsigned wc = wcoord[ch];
        const int n = n0 + (int)(wc & 255u);
        const int it = vt0 + (int)((wc >> 8) & 255u);
        const int ih = vh0 + (int)((wc >> 16) & 255u);
        const int iw = vw0 + (int)(wc >> 24);
        const bool ok = wc != 0xffffffffu && n < a.N && it >= 0 && ih >= 0 && iw >= 0 &&
                        it < a.Ti && ih < a.Hi && iw < a.Wi;
        const float* base = a.x + (long)n * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw +
                            (long)cin_lo * a.x_cstride;
        for (int c = wave; c < nci; c += 4) {
          const float* src = ok ? base + (long)c * a.x_cstride : g_zero;
          GLDS4(src, Xs + c * planeP + ch * 64);
        }
      }
    }
    __syncthreads();     // drains the DMA queue (vmcnt(0)) and publishes the tile

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


// ------------------------------------------------------------------------------------
// Second-generation kernel: loader / compute wave specialisation.
//
//   * lanes of the B operand are input CHANNELS (row stride planeP is odd -> every
//     ds_read_b32 is conflict free); the stencil taps are a register loop: one wave owns
//     a 32co x 32ci x TAPS block of dW (TAPS accumulators) and reuses each dY operand for
//     all taps.  (The first-generation kernel above maps lanes to (ci,tap) pairs and loses
//     ~70 % of its LDS cycles to bank conflicts; it stays as the fallback for Cin = 3 and
//     windows that do not fit two LDS stages.)
//   * 512 threads (768 for the temporal forms, NL = 8): waves 0-3 only issue MFMAs, waves 4..3+NL
//     only run the LDS-DMA engine
//     (buffer_load_dword ... lds through bounds-checked descriptors: padding / overhang
//     lanes land as 0.0).  Two LDS stages, ONE barrier per 64-position box: the loaders
//     fill box b+1 while the matrix waves consume box b.
//   * the position -> window offset of MFMA step s is wave-uniform, so it is scalar ALU
//     work; the per-read VALU cost is a single v_add.
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
constexpr unsigned W2_OOB = 0x80000000u;

struct Wgrad2Args {
  const float* x;
  const float* dy;
  float* part;            // [S][Cout][J]
  long x_nstride, dy_nstride;
  int x_cstride, dy_cstride;
  int N, Cin, Cout, J;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw;
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int WT, WH, WW, plane1, plane, planeP;
  float inv_plane1, inv_hw, inv_ww;
  int ntiles, S;
  int To_full;            // Winograd form: a.To counts frame PAIRS, this is the real frame count
  int xcd;                // XCD-aware (split, tile) ids, see wgrad_tile
  // stem kernel, BNA form: `dy` is d(activation) of the BatchNorm(+ReLU) unit behind the convolution; the operand
  // the matrix waves multiply is A[co] * g + B[co] * y + D[co], g = dz masked by (y * scale + shift > 0)
  const float* bn_y;      // the convolution's own output
  const float* bn_coef;   // [5][Cout]: A, B, D, scale, shift
  long bn_y_nstride;
  int bn_relu;
};

// (split, ci tile, co tile) of this workgroup.  Plain: the launch grid's.  XCD-aware (a.xcd): the hardware
// deals workgroups in linear order (x fastest) round-robin over the 8 XCDs; the linear id is remapped so
// that XCD x owns a contiguous run of logical ids, and logical ids run TILE-fastest: the ct * mt workgroups
// that stream the same boxes (one split) are neighbours in time on one XCD and share its L2 -- with the
// split fastest they were a whole dispatch wave apart and every tile fetched both operands from HBM
// (rocprofv3 FETCH_SIZE on Conv_2c.conv2: 3.3x the two tensors).
struct WgTile { int split, ct, mt; };
__device__ __forceinline__ WgTile wgrad_tile(int xcd) {
  WgTile t;
  t.split = blockIdx.x; t.ct = blockIdx.y; t.mt = blockIdx.z;
  if (xcd) {
    const int tiles = (int)(gridDim.y * gridDim.z), total = (int)gridDim.x * tiles;
    int lin = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int per = total >> 3;
    if (lin < (per << 3)) lin = (lin & 7) * per + (lin >> 3);
    t.split = lin / tiles;
    const int tile = lin - t.split * tiles;
    t.mt = tile / (int)gridDim.y;
    t.ct = tile - t.mt * (int)gridDim.y;
  }
  return t;
}

__device__ __forceinline__ int w2_fdiv(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

//
// WINO (temporal (3,1,1) stride-1 pad-1 stencils): Winograd F(2,3) along T, the form the forward
// and data-gradient passes of these layers already use.  A box holds 32 frame PAIRS; per pair
//     dU0 += dy0 (d0-d2)   dU1 += (dy0+dy1)(d1+d2)   dU2 += (dy0-dy1)(d2-d1)   dU3 += dy1 (d3-d1)
// (d0..d3 = input frames 2p-1..2p+2): FOUR MFMAs per pair instead of the six of two direct
// positions; the epilogue maps back dg0 = dU0 + (dU1+dU2)/2, dg1 = (dU1-dU2)/2,
// dg2 = (dU1+dU2)/2 + dU3.  Seen from the code below it is a (4,1,1) stencil with temporal stride 2
// over pair positions whose operands are sums / differences of two LDS reads.
//
// WHW ((1,3,3) stride-1 pad-1 stencils on even maps, the layers whose forward runs through F(2x2,3x3)):
// the gradient is accumulated in the Winograd domain.  Per 2x2 block of outputs
//     dM = A dY A^T (4x4 from the 2x2 block of dY)      V = B^T d B (4x4 from the 4x4 input patch)
//     dU[xi] (co, ci) += dM[xi] (co, block) * V[xi] (ci, block)          xi = 4i + j
// and the epilogue maps back dg = G^T dU G: SIXTEEN MFMAs per two blocks (8 positions) instead of the
// 36 of the direct form.  A workgroup owns 32 co x 64 ci; matrix wave (xh, wn) owns rows i = 2xh, 2xh+1 of
// xi for ci block wn: 8 accumulator sets (128 registers, so a loader wave still shares its SIMD).  G^T . G
// is linear in dU, so each xi-half writes its own partial 3x3 as an extra split slice (2 S slices in all)
// and wgrad_reduce_kernel folds them.  Signs: A = [[1,0],[1,1],[1,-1],[0,-1]]; the kernel feeds +b where
// A says -b (row 3 / column 3 of dM) and the epilogue applies s(i) s(j), s = (+,+,+,-).
template <int KT, int KH, int KW, int MB, int NB, int PCH, bool WINO = false, int NL = 4, bool WHW = false>
__global__ void __launch_bounds__(256 + 64 * NL)
conv_wgrad2_kernel(const Wgrad2Args a) {
  constexpr int TAPS = KT * KH * KW;
  constexpr int BP = WINO ? 32 : 64;     // positions (WINO: frame pairs) per box
  constexpr int BMt = WHW ? 32 : 64 * MB, BCt = 64 * NB;
  static_assert(!WHW || (KT == 1 && KH == 3 && KW == 3 && MB == 1 && NB == 1 && !WINO), "F(2x2,3x3) form");
  constexpr int LDY = 64 + 1;            // WINO: [32 first frames | 32 second frames] per row
  constexpr int STEPS = BP / 2;
  static_assert(!WINO || (KT == 4 && KH == 1 && KW == 1), "Winograd form is the (4,1,1)/2 view");

  extern __shared__ __align__(16) float smem[];
  const int planeP = a.planeP;
  const int stage_floats = BMt * LDY + BCt * planeP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgTile wt = wgrad_tile(a.xcd);
  const int split = wt.split;
  const int ci0 = wt.ct * BCt, co0 = wt.mt * BMt;
  const int nbox = (a.ntiles - split + a.S - 1) / a.S;
  const int lW = a.lTW, lWH = a.lTW + a.lTH, lWHT = a.lTW + a.lTH + a.lTT;

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const int lw = wave - 4;
    // box-independent decode of the elements this lane moves
    int wn_[PCH], wt_[PCH], wh_[PCH], ww_[PCH];
    {
      const int hw = a.WH * a.WW;
#pragma unroll
      for (int j = 0; j < PCH; ++j) {
        const int e = j * 64 + lane;
        const int q0 = w2_fdiv(e, a.inv_plane1);
        int q = e - q0 * a.plane1;
        const int t = w2_fdiv(q, a.inv_hw); q -= t * hw;
        const int h = w2_fdiv(q, a.inv_ww);
        wn_[j] = q0; wt_[j] = t; wh_[j] = h; ww_[j] = q - h * a.WW;
      }
    }
    const int plane_ = WINO ? (lane & 31) : lane;      // WINO: lanes 32-63 fetch the pair's 2nd frame
    const int pfr = WINO ? (lane >> 5) : 0;
    const int ptw = plane_ & ((1 << lW) - 1);
    const int pth = (plane_ >> lW) & ((1 << a.lTH) - 1);
    const int ptt = (plane_ >> lWH) & ((1 << a.lTT) - 1);
    const int ptn = plane_ >> lWHT;

    for (int b = 0; b < nbox; ++b) {
      const int tile = split + b * a.S;
      int r = tile;
      const int bw_ = r % a.nbw; r /= a.nbw;
      const int bh_ = r % a.nbh; r /= a.nbh;
      const int bt_ = r % a.nbt; r /= a.nbt;
      const int n0 = r << a.lTN;
      const int ow0 = bw_ << lW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;
      float* dYs = smem + (b & 1) * stage_floats;
      float* Xs = dYs + BMt * LDY;

      // ---- dY[BMt][64]: rows lw, lw+4, ... ----------------------------------------
      {
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.dy + (long)n0 * a.dy_nstride), 0, 0x80000000u, 0x00020000);
        const int n = n0 + ptn, oh = oh0 + pth, ow = ow0 + ptw;
        const int ot = WINO ? 2 * (ot0 + ptt) + pfr : ot0 + ptt;          // real output frame
        const bool ok = n < a.N && ot < (WINO ? a.To_full : a.To) && oh < a.Ho && ow < a.Wo;
        const unsigned voff =
            ok ? (unsigned)(((long)ptn * a.dy_nstride + ((long)ot * a.Ho + oh) * a.Wo + ow) * 4)
               : W2_OOB;
        for (int row = lw; row < BMt; row += NL) {
          const int co = co0 + row;
          if (co < a.Cout)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS_PTR(dYs + row * LDY), 4, voff,
                                                     (unsigned)co * (unsigned)a.dy_cstride * 4u,
                                                     0, 0);
          else if (b < 2)
            dYs[row * LDY + lane] = 0.f;
        }
      }
      // ---- X window: channels lw, lw+4, ... ----------------------------------------
      {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.x + (long)n0 * a.x_nstride), 0, 0x80000000u, 0x00020000);
        const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
        unsigned voff[PCH];
#pragma unroll
        for (int j = 0; j < PCH; ++j) {
          const int n = n0 + wn_[j], it = vt0 + wt_[j], ih = vh0 + wh_[j], iw = vw0 + ww_[j];
          const bool ok = n < a.N && it >= 0 && ih >= 0 && iw >= 0 && it < a.Ti && ih < a.Hi &&
                          iw < a.Wi;
          voff[j] = ok ? (unsigned)(((long)wn_[j] * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi +
                                     iw) * 4)
                       : W2_OOB;
        }
#ifdef COCLR_WGRAD_ABLATE
        // timing ablation (wrong results by design; tools/wgrad_ablate.sh): no window DMA after the first
        // two boxes.  Measured at B=32: Conv_2c.conv1 (F(2x2,3x3) form) 0.713 -> 0.704 ms, Conv_2c.conv2
        // (F(2,3) form) 0.883 -> 0.815, Conv_1a.conv2 1.038 -> 1.016: the loader waves do not pace these kernels
        if (b >= 2) goto loaded;
#endif
        for (int c = lw; c < BCt; c += NL) {
          const int ci = ci0 + c;
          if (ci < a.Cin) {
            const unsigned soff = (unsigned)ci * (unsigned)a.x_cstride * 4u;
#pragma unroll
            for (int j = 0; j < PCH; ++j)
              if (j * 64 + lane < a.plane)     // exec-masked: rows are packed at planeP
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(Xs + c * planeP + j * 64), 4,
                                                         voff[j], soff, 0, 0);
          } else if (b < 2) {
#pragma unroll
            for (int j = 0; j < PCH; ++j)
              if (j * 64 + lane < a.plane) Xs[c * planeP + j * 64 + lane] = 0.f;
          }
        }
      }
#ifdef COCLR_WGRAD_ABLATE
      loaded:;
#endif
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's share of box b is in LDS
      __syncthreads();                      // barrier b
    }
    return;
  }

  // ================================ matrix waves ================================
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  if constexpr (WHW) {
    const int xh = wm;                                   // rows i = 2 xh, 2 xh + 1 of xi = 4 i + j
    const int TWb = 1 << lW;                             // box width in positions
    const int abase = l31 * LDY + 2 * half;              // lanes 32-63: the block two columns to the right
    const int jbase = BMt * LDY + (wn * 32 + l31) * planeP + 2 * half;
    f32x16 accw[2][4];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accw[ii][j][e] = 0.f;
    // block pair s: blocks 2s, 2s+1 are neighbours along w (box width >= 4); wave-uniform decode
    auto offs = [&](int s, int& p, int& wo) {
      const int b0 = 2 * s;
      const int bw = b0 & ((1 << (lW - 1)) - 1);
      int r = b0 >> (lW - 1);
      const int bh = r & ((1 << (a.lTH - 1)) - 1); r >>= (a.lTH - 1);
      const int tt = r & ((1 << a.lTT) - 1);
      const int tn = r >> a.lTT;
      p = ((((tn << a.lTT) | tt) << a.lTH | (2 * bh)) << lW) | (2 * bw);
      wo = tn * a.plane1 + (tt * a.WH + 2 * bh) * a.WW + 2 * bw;
    };
    for (int b = 0; b < nbox; ++b) {
      const float* cur = smem + (b & 1) * stage_floats;
      __syncthreads();   // barrier b
      float ya[2][4], dv[2][3][4];
      auto fetch = [&](int s, float (&Y)[4], float (&D)[3][4]) {
        int p, wo;
        offs(s, p, wo);
        const float* yp = cur + abase + p;
        Y[0] = yp[0]; Y[1] = yp[1]; Y[2] = yp[TWb]; Y[3] = yp[TWb + 1];
        const float* xp = cur + jbase + wo + xh * a.WW;          // patch rows xh .. xh + 2
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) D[r][c] = xp[r * a.WW + c];
      };
      fetch(0, ya[0], dv[0]);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 1 < 8) fetch(s + 1, ya[(s + 1) & 1], dv[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const float (&Y)[4] = ya[s & 1];
        const float (&D)[3][4] = dv[s & 1];
        // dM rows of this half (signs deferred): xh = 0: y_row0, y_row0 + y_row1;  xh = 1: y_row0 - y_row1, y_row1
        float ra[2][2];
        if (xh == 0) { ra[0][0] = Y[0]; ra[0][1] = Y[1]; ra[1][0] = Y[0] + Y[2]; ra[1][1] = Y[1] + Y[3]; }
        else         { ra[0][0] = Y[0] - Y[2]; ra[0][1] = Y[1] - Y[3]; ra[1][0] = Y[2]; ra[1][1] = Y[3]; }
        // B^T d rows of this half: xh = 0: d0 - d2, d1 + d2 (patch rows 0,1,2);  xh = 1: d2 - d1, d1 - d3 (rows 1,2,3)
        float rb[2][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (xh == 0) { rb[0][c] = D[0][c] - D[2][c]; rb[1][c] = D[1][c] + D[2][c]; }
          else         { rb[0][c] = D[1][c] - D[0][c]; rb[1][c] = D[0][c] - D[2][c]; }
        }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const float a0 = ra[ii][0], a1 = ra[ii][1];
          const float A4[4] = {a0, a0 + a1, a0 - a1, a1};
          const float B4[4] = {rb[ii][0] - rb[ii][2], rb[ii][1] + rb[ii][2], rb[ii][2] - rb[ii][1],
                               rb[ii][1] - rb[ii][3]};
#pragma unroll
          for (int j = 0; j < 4; ++j)
            accw[ii][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(A4[j], B4[j], accw[ii][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // partial dg of this xi-half: dg = G^T dU G restricted to its two rows; slice 2 * split + xh
    float* out = a.part + (long)(2 * split + xh) * a.Cout * a.J;
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * half;
      if (co < a.Cout && ci < a.Cin) {
        float t[2][3];                                   // column pass (over j), s(3) = -1
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const float u0 = accw[ii][0][e], u1 = accw[ii][1][e], u2 = accw[ii][2][e], u3 = accw[ii][3][e];
          const float hs = 0.5f * (u1 + u2);
          t[ii][0] = u0 + hs;
          t[ii][1] = 0.5f * (u1 - u2);
          t[ii][2] = hs - u3;
        }
        float* dst = out + (long)co * a.J + ci * 9;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          // row pass (over i): rows 0,1 (xh = 0): g0 = t0 + t1/2, g1 = t1/2, g2 = t1/2
          //                    rows 2,3 (xh = 1, row 3 negated): g0 = t2/2, g1 = -t2/2, g2 = t2/2 - t3
          const float h1 = 0.5f * (xh == 0 ? t[1][kw] : t[0][kw]);
          if (xh == 0) { dst[kw] = t[0][kw] + h1; dst[3 + kw] = h1; dst[6 + kw] = h1; }
          else         { dst[kw] = h1; dst[3 + kw] = -h1; dst[6 + kw] = h1 - t[1][kw]; }
        }
      }
    }
    return;
  }
  int abase[MB], jb[NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) abase[mb] = ((wm * MB + mb) * 32 + l31) * LDY + half;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    jb[nb] = BMt * LDY + ((wn * NB + nb) * 32 + l31) * planeP + half * a.sw;

  f32x16 acc[MB][NB][TAPS];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mb][nb][t][i] = 0.f;

  // window offset of position 2s (wave-uniform: scalar ALU)
  auto wo_of = [&](int s) {
    const int p0 = 2 * s;
    const int tw = p0 & ((1 << lW) - 1);
    const int th = (p0 >> lW) & ((1 << a.lTH) - 1);
    const int tt = (p0 >> lWH) & ((1 << a.lTT) - 1);
    const int tn = p0 >> lWHT;
    return tn * a.plane1 + ((tt * a.st) * a.WH + th * a.sh) * a.WW + tw * a.sw;
  };

  for (int b = 0; b < nbox; ++b) {
    const float* cur = smem + (b & 1) * stage_floats;
    __syncthreads();   // barrier b: box b is in LDS, box b-1 fully consumed

    float av[2][MB], av1[2][MB], bv[2][NB][TAPS];
    auto fetch = [&](int s, float (&A)[MB], float (&A1)[MB], float (&B)[NB][TAPS]) {
      const int wo = wo_of(s);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        A[mb] = cur[abase[mb] + 2 * s];
        if (WINO) A1[mb] = cur[abase[mb] + 32 + 2 * s];      // second frame of the pair
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int xb = jb[nb] + wo;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
          const int kt = t / (KH * KW), kh = (t / KW) % KH, kw = t % KW;
          B[nb][t] = cur[xb + (kt * a.WH + kh) * a.WW + kw];
        }
      }
    };
    fetch(0, av[0], av1[0], bv[0]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + 1 < STEPS) fetch(s + 1, av[(s + 1) & 1], av1[(s + 1) & 1], bv[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      if (WINO) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const float y0 = av[s & 1][mb], y1 = av1[s & 1][mb];
          const float A4[4] = {y0, y0 + y1, y0 - y1, y1};
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) {
            const float d0 = bv[s & 1][nb][0], d1 = bv[s & 1][nb][1], d2 = bv[s & 1][nb][2],
                        d3 = bv[s & 1][nb][3];
            const float B4[4] = {d0 - d2, d1 + d2, d2 - d1, d3 - d1};
#pragma unroll
            for (int t = 0; t < 4; ++t)
              acc[mb][nb][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(A4[t], B4[t], acc[mb][nb][t],
                                                                   0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                  av[s & 1][mb], bv[s & 1][nb][t], acc[mb][nb][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // partial tile out: part[split][co][ci*TAPS + t]
  float* out = a.part + (long)split * a.Cout * a.J;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + (wm * MB + mb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (co < a.Cout) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int ci = ci0 + (wn * NB + nb) * 32 + l31;
          if (ci < a.Cin) {
            if (WINO) {
              float* dst = out + (long)co * a.J + ci * 3;        // back to the three real taps
              const float u0 = acc[mb][nb][0][i], u1 = acc[mb][nb][1][i], u2 = acc[mb][nb][2][i],
                          u3 = acc[mb][nb][3][i];
              const float hs = 0.5f * (u1 + u2);
              dst[0] = u0 + hs;
              dst[1] = 0.5f * (u1 - u2);
              dst[2] = hs + u3;
            } else {
              float* dst = out + (long)co * a.J + ci * TAPS;
#pragma unroll
              for (int t = 0; t < TAPS; ++t) dst[t] = acc[mb][nb][t][i];
            }
          }
        }
      }
    }
}

// ------------------------------------------------------------------------------------
// Pointwise (1x1x1) weight gradient: dW[co][ci] = sum_p dY[co][p] * X[ci][p] is a plain GEMM
// whose two operands are both runs of contiguous positions.  The generic v2 kernel moves them
// with 4-byte DMA (256 instructions per 128x128x64 box: the loader waves, not the matrix
// pipes, set its pace -- 38 TFLOP/s).  Here
//   * both tiles are [row][64 positions] with NO padding, filled by 16-byte LDS-DMA
//     (64 instructions per box);
//   * bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index with the row
//     (applied on the SOURCE side of the DMA, whose LDS destination is fixed per lane), and the
//     operands are fetched with ds_read_b128: lane (row, half) gets positions 8j+4*half..+3 and
//     feeds them to four consecutive MFMA steps (step t pairs positions 8j+t and 8j+4+t for
//     both operands), i.e. one LDS read per operand block per 4 steps.
// Requirements (checked on the host): stride 1, plane size a multiple of 64, 16-byte aligned
// tensors and strides.
template <int MB, int NB>
__global__ void __launch_bounds__(512)
conv_wgrad_pw_kernel(const Wgrad2Args a) {
  constexpr int BMt = 64 * MB, BCt = 64 * NB;
  constexpr int ROW = 64;                               // floats per row = positions per box
  constexpr int stage_floats = (BMt + BCt) * ROW;
  constexpr int APIECES = BMt / 4, BPIECES = BCt / 4;  // 1 KiB pieces (4 rows each)

  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const WgTile wt = wgrad_tile(a.xcd);
  const int split = wt.split;
  const int ci0 = wt.ct * BCt, co0 = wt.mt * BMt;
  const int nbox = (a.ntiles - split + a.S - 1) / a.S;
  const int S = a.Wi;                                   // flattened plane size
  const int boxes_per_sample = S >> 6;

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const int lw = wave - 4;
    const int r = lane >> 4, q = lane & 15;             // row inside a piece, LDS chunk slot
    unsigned voff_a[4], voff_b[4];                      // by (piece & 3): source chunk = q ^ (row & 15)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c4 = q ^ ((k * 4 + r) & 15);
      voff_a[k] = (unsigned)(((long)r * a.dy_cstride + c4 * 4) * 4);
      voff_b[k] = (unsigned)(((long)r * a.x_cstride + c4 * 4) * 4);
    }
    for (int b = 0; b < nbox; ++b) {
      const int tile = split + b * a.S;
      const int n = tile / boxes_per_sample;
      const int p0 = (tile - n * boxes_per_sample) << 6;
      float* As = smem + (b & 1) * stage_floats;
      float* Bs = As + BMt * ROW;
      const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.dy + (long)n * a.dy_nstride + p0), 0, 0x80000000u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.x + (long)n * a.x_nstride + p0), 0, 0x80000000u, 0x00020000);
      for (int p = lw; p < APIECES; p += 4) {
        const int row = co0 + p * 4;
        const unsigned vo = row + r < a.Cout ? voff_a[p & 3] : W2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS_PTR(As + p * 256), 16, vo,
                                                 (unsigned)row * (unsigned)a.dy_cstride * 4u, 0, 0);
      }
      for (int p = lw; p < BPIECES; p += 4) {
        const int row = ci0 + p * 4;
        const unsigned vo = row + r < a.Cin ? voff_b[p & 3] : W2_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(Bs + p * 256), 16, vo,
                                                 (unsigned)row * (unsigned)a.x_cstride * 4u, 0, 0);
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      __syncthreads();                      // barrier b
    }
    return;
  }

  // ================================ matrix waves ================================
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  int arow[MB], brow[NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) arow[mb] = (wm * MB + mb) * 32 + l31;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) brow[nb] = (wn * NB + nb) * 32 + l31;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  for (int b = 0; b < nbox; ++b) {
    const float* As = smem + (b & 1) * stage_floats;
    const float* Bs = As + BMt * ROW;
    __syncthreads();   // barrier b
    f32x4 av[2][MB], bv[2][NB];
    auto fetch = [&](int j, f32x4 (&A)[MB], f32x4 (&B)[NB]) {
      const int c4 = 2 * j + half;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        A[mb] = *reinterpret_cast<const f32x4*>(As + arow[mb] * ROW + ((c4 ^ (arow[mb] & 15)) << 2));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        B[nb] = *reinterpret_cast<const f32x4*>(Bs + brow[nb] * ROW + ((c4 ^ (brow[nb] & 15)) << 2));
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (j + 1 < 8) fetch(j + 1, av[(j + 1) & 1], bv[(j + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j & 1][mb][t], bv[j & 1][nb][t],
                                                               acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float* out = a.part + (long)split * a.Cout * a.J;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + (wm * MB + mb) * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (co < a.Cout) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int ci = ci0 + (wn * NB + nb) * 32 + l31;
          if (ci < a.Cin) out[(long)co * a.J + ci] = acc[mb][nb][i];
        }
      }
    }
}

// ------------------------------------------------------------------------------------
// Stem weight gradient: (1,KH,KW) stencil over CIN = 3 input channels (s3dg.py:145 conv1 and
// the slices of resnet_2d3d.py:138).  J = CIN*KH*KW = 147 columns would half-fill a second
// 128-column tile of the (ci,tap)-lane kernel and its window reads collide in LDS.  Here:
//   * every matrix wave owns the WHOLE 64 x 160 (= J padded to 5 x 32) tile of dW and a
//     quarter of each box's positions (split-K inside the workgroup: the four partial tiles
//     go out as four split slices, folded by wgrad_reduce_kernel with the rest);
//   * per MFMA step: 2 dY operands + 5 window operands -> 10 MFMAs;
//   * the window sits in LDS with a row pitch == 7 (mod 32), so the 32 (kh,kw) offsets of a
//     column block fall into distinct banks;
//   * waves 4-7 are loaders (LDS-DMA, two stages, one barrier per 128-position box).
//   * BNA: the BatchNorm backward apply pass of the unit behind the convolution happens HERE.  Nobody
//     else reads d(conv output) of a stem (the clip needs no gradient), so the 1 GB tensor at B = 32 is
//     neither written nor re-read: the loaders bring dz AND y rows plus one validity flag per position,
//     the matrix waves form fmaf(A, g, fmaf(B, y, D)) -- the apply kernel's own expression, so the
//     result is bit-identical to apply-then-multiply -- on their way from LDS to the MFMA.
template <int KH, int KW, int CIN, int PCH, bool BNA = false>
__global__ void __launch_bounds__(512)
conv_wgrad_stem_kernel(const Wgrad2Args a) {
  constexpr int TAPS = KH * KW, J = CIN * TAPS, NJB = (J + 31) / 32;
  constexpr int BP = 128;                // positions per box
  constexpr int BMt = 64;
  constexpr int LDY = BP + 1;
  constexpr int STEPS = BP / 8;          // per matrix wave: BP/4 positions, 2 per step
  constexpr int YOFF = BMt * LDY;                          // BNA: y rows behind the dz rows
  constexpr int VOFF = 2 * BMt * LDY;                      // BNA: BP validity flags behind them
  constexpr int XOFF = BNA ? 2 * BMt * LDY + BP : BMt * LDY;

  extern __shared__ __align__(16) float smem[];
  const int planeP = a.planeP;
  const int stage_floats = XOFF + CIN * planeP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x;
  const int co0 = blockIdx.z * BMt;
  const int nbox = (a.ntiles - split + a.S - 1) / a.S;
  const int lW = a.lTW, lWH = a.lTW + a.lTH, lWHT = a.lTW + a.lTH + a.lTT;

  if (wave >= 4) {
    // =============================== loader waves ===============================
    const int lw = wave - 4;
    constexpr int PW = (PCH + 3) / 4;    // window pieces lw, lw+4, ... of every channel
    int wn_[PW], wt_[PW], wh_[PW], ww_[PW];
    {
      const int hw = a.WH * a.WW;
#pragma unroll
      for (int jj = 0; jj < PW; ++jj) {
        const int e = (lw + 4 * jj) * 64 + lane;
        const int q0 = w2_fdiv(e, a.inv_plane1);
        int q = e - q0 * a.plane1;
        const int t = w2_fdiv(q, a.inv_hw); q -= t * hw;
        const int h = w2_fdiv(q, a.inv_ww);
        wn_[jj] = q0; wt_[jj] = t; wh_[jj] = h; ww_[jj] = q - h * a.WW;
      }
    }
    for (int b = 0; b < nbox; ++b) {
      const int tile = split + b * a.S;
      int r = tile;
      const int bw_ = r % a.nbw; r /= a.nbw;
      const int bh_ = r % a.nbh; r /= a.nbh;
      const int bt_ = r % a.nbt; r /= a.nbt;
      const int n0 = r << a.lTN;
      const int ow0 = bw_ << lW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;
      float* dYs = smem + (b & 1) * stage_floats;
      float* Xs = dYs + XOFF;
      // ---- dY[64][128]: rows lw, lw+4, ...; two 64-position pieces per row --------------
      {
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.dy + (long)n0 * a.dy_nstride), 0, 0x80000000u, 0x00020000);
        unsigned voff[2], yoff[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = h * 64 + lane;
          const int tw = p & ((1 << lW) - 1);
          const int th = (p >> lW) & ((1 << a.lTH) - 1);
          const int tt = (p >> lWH) & ((1 << a.lTT) - 1);
          const int tn = p >> lWHT;
          const int n = n0 + tn, ot = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
          const bool ok = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
          const long sp = ((long)ot * a.Ho + oh) * a.Wo + ow;
          voff[h] = ok ? (unsigned)(((long)tn * a.dy_nstride + sp) * 4) : W2_OOB;
          if (BNA) {
            yoff[h] = ok ? (unsigned)(((long)tn * a.bn_y_nstride + sp) * 4) : W2_OOB;
            if (lw == 0) dYs[VOFF + p] = ok ? 1.f : 0.f;
          }
        }
        for (int row = lw; row < BMt; row += 4) {
          const int co = co0 + row;
          const unsigned soff = (unsigned)co * (unsigned)a.dy_cstride * 4u;
          if (co < a.Cout) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, LDS_PTR(dYs + row * LDY + h * 64), 4,
                                                       voff[h], soff, 0, 0);
          } else if (b < 2) {
            dYs[row * LDY + lane] = 0.f;
            dYs[row * LDY + 64 + lane] = 0.f;
          }
        }
        if (BNA) {
          const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.bn_y + (long)n0 * a.bn_y_nstride), 0, 0x80000000u, 0x00020000);
          for (int row = lw; row < BMt; row += 4) {
            const int co = co0 + row;
            const unsigned soff = (unsigned)co * (unsigned)a.dy_cstride * 4u;
            if (co < a.Cout) {
#pragma unroll
              for (int h = 0; h < 2; ++h)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, LDS_PTR(dYs + YOFF + row * LDY + h * 64), 4,
                                                         yoff[h], soff, 0, 0);
            } else if (b < 2) {
              dYs[YOFF + row * LDY + lane] = 0.f;
              dYs[YOFF + row * LDY + 64 + lane] = 0.f;
            }
          }
        }
      }
      // ---- X window ---------------------------------------------------------------------
      {
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.x + (long)n0 * a.x_nstride), 0, 0x80000000u, 0x00020000);
        const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
#pragma unroll
        for (int jj = 0; jj < PW; ++jj) {
          const int j = lw + 4 * jj;
          if (j * 64 + lane < a.plane) {     // exec-masked: rows are packed at planeP
            const int n = n0 + wn_[jj], it = vt0 + wt_[jj], ih = vh0 + wh_[jj], iw = vw0 + ww_[jj];
            const bool ok = n < a.N && it >= 0 && ih >= 0 && iw >= 0 && it < a.Ti && ih < a.Hi &&
                            iw < a.Wi;
            const unsigned voff =
                ok ? (unsigned)(((long)wn_[jj] * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4)
                   : W2_OOB;
#pragma unroll
            for (int c = 0; c < CIN; ++c)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(Xs + c * planeP + j * 64), 4, voff,
                                                       (unsigned)c * (unsigned)a.x_cstride * 4u, 0, 0);
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
      __syncthreads();                      // barrier b
    }
    return;
  }

  // ================================ matrix waves ================================
  const int half = lane >> 5, l31 = lane & 31;
  int abase[2], jb[NJB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) abase[mb] = (mb * 32 + l31) * LDY + wave * (BP / 4) + half;
#pragma unroll
  for (int nb = 0; nb < NJB; ++nb) {
    int j = nb * 32 + l31;
    if (j >= J) j = J - 1;                 // pad columns: computed, never stored
    const int c = j / TAPS, tap = j - c * TAPS;
    jb[nb] = XOFF + c * planeP + (tap / KW) * a.WW + (tap % KW) + half * a.sw;
  }
  float cA[2], cB[2], cD[2], cS[2], cF[2];
  if (BNA) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int co = co0 + mb * 32 + l31;
      const bool in = co < a.Cout;
      cA[mb] = in ? a.bn_coef[co] : 0.f;
      cB[mb] = in ? a.bn_coef[a.Cout + co] : 0.f;
      cD[mb] = in ? a.bn_coef[2 * a.Cout + co] : 0.f;
      cS[mb] = in ? a.bn_coef[3 * a.Cout + co] : 0.f;
      cF[mb] = in ? a.bn_coef[4 * a.Cout + co] : 0.f;
    }
  }
  const int vbase = VOFF + wave * (BP / 4) + half;
  const bool bn_relu = BNA && a.bn_relu;
  f32x16 acc[2][NJB];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < NJB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][nb][i] = 0.f;

  auto wo_of = [&](int s) {              // window offset of this wave's position 2s
    const int p0 = wave * (BP / 4) + 2 * s;
    const int tw = p0 & ((1 << lW) - 1);
    const int th = (p0 >> lW) & ((1 << a.lTH) - 1);
    const int tt = (p0 >> lWH) & ((1 << a.lTT) - 1);
    const int tn = p0 >> lWHT;
    return tn * a.plane1 + ((tt * a.st) * a.WH + th * a.sh) * a.WW + tw * a.sw;
  };

  for (int b = 0; b < nbox; ++b) {
    const float* cur = smem + (b & 1) * stage_floats;
    __syncthreads();   // barrier b
    float av[2][2], bv[2][NJB];
    auto fetch = [&](int s, float (&A)[2], float (&B)[NJB]) {
      const int wo = wo_of(s);
      if (BNA) {
        const float vm = cur[vbase + 2 * s];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          float g = cur[abase[mb] + 2 * s];
          const float v = cur[YOFF + abase[mb] + 2 * s];
          if (bn_relu) g = fmaf(v, cS[mb], cF[mb]) > 0.f ? g : 0.f;
          A[mb] = fmaf(cA[mb], g, fmaf(cB[mb], v, cD[mb] * vm));
        }
      } else {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) A[mb] = cur[abase[mb] + 2 * s];
      }
#pragma unroll
      for (int nb = 0; nb < NJB; ++nb) B[nb] = cur[jb[nb] + wo];
    };
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      if (s + 1 < STEPS) fetch(s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NJB; ++nb)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][mb], bv[s & 1][nb],
                                                             acc[mb][nb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // partial tile of this wave: slice (split*4 + wave) of part[S*4][Cout][J]
  float* out = a.part + ((long)split * 4 + wave) * a.Cout * J;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = co0 + mb * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (co < a.Cout) {
#pragma unroll
        for (int nb = 0; nb < NJB; ++nb) {
          const int j = nb * 32 + l31;
          if (j < J) out[(long)co * J + j] = acc[mb][nb][i];
        }
      }
    }
}

// dW[co][ci*ci_stride' ...] = sum_s part[s][e]; destination may be a tap slice of a
// larger stencil (r50 stem): e = co*J + ci*taps + tap ->
// dst[co*co_stride + ci*ci_stride + tap_base + tap].
// Rows [end[i-1], end[i]) of the reduced matrix land in dst[i] (row index local to the segment): the
// fused 1x1x1 heads of an inception block are ONE weight-gradient GEMM whose row blocks belong to
// three parameters living at unrelated addresses (views of DistributedDataParallel's buckets).
struct WgradDst {
  float* p[4];
  int end[4];
  int n;
};

__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ part, WgradDst dst, long CJ, int S, int J,
                    int taps, long co_stride, long ci_stride, int tap_base, int accumulate) {
  // 64 consecutive elements x 4 split groups per block; 4 independent loads in flight per
  // thread, fixed summation order (deterministic)
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long e0 = (long)blockIdx.x * 64; e0 < CJ; e0 += (long)gridDim.x * 64) {
    const long e = e0 + tx;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < CJ) {
      int k = ty;
      for (; k + 12 < S; k += 16) {
        s0 += part[(long)k * CJ + e];
        s1 += part[(long)(k + 4) * CJ + e];
        s2 += part[(long)(k + 8) * CJ + e];
        s3 += part[(long)(k + 12) * CJ + e];
      }
      for (; k < S; k += 4) s0 += part[(long)k * CJ + e];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && e < CJ) {
      const float s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
      long co = e / J;
      const int j = (int)(e - co * J);
      const int ci = j / taps, tap = j - ci * taps;
      int sg = 0;
      while (sg < dst.n - 1 && co >= dst.end[sg]) ++sg;
      if (sg) co -= dst.end[sg - 1];
      float* d = dst.p[sg] + co * co_stride + ci * ci_stride + tap_base + tap;
      *d = accumulate ? *d + s : s;
    }
    __syncthreads();
  }
}

struct WPlan {
  ConvPlan p;
  int variant, BJ, S, jtiles, mtiles, planeP, pch, nci_max;
  // second-generation kernel
  int v2;            // 0: not applicable, else variant id (6: temporal Winograd form of id 2)
  int pw;            // pointwise: 16-byte-DMA GEMM kernel applicable
  ConvPlan p2;
  int S2, planeP2, mt2, ct2;
  size_t lds2;
};

// Split count as a multiple of 8: the grid is (split, ci tile, co tile) with the split fastest, and the
// hardware deals consecutive workgroup ids round-robin over the 8 XCDs -- with S % 8 == 0 the workgroups
// that stream the SAME boxes (same split, other tiles) sit on one XCD and share its L2.
inline int xcd_round(int S) {
  static const bool off = getenv("COCLR_XCD_MAP") && atoi(getenv("COCLR_XCD_MAP")) == 0;
  return (!off && S >= 16) ? (S & ~7) : S;
}

// Order of the (split, ci tile, co tile) workgroups of the second-generation / pointwise kernels, see
// wgrad_tile(): 1 = tile-fastest XCD-aware ids, 0 = the launch grid's split-fastest order.
// COCLR_WGRAD_ORDER=split|tile forces one (read per call: tools/wgrad_order_ab.py alternates in-process).
inline int wgrad_order_env() {
  const char* e = getenv("COCLR_WGRAD_ORDER");
  if (!e) return -1;
  return e[0] == 't' ? 1 : (e[0] == 's' ? 0 : -1);
}

// tile of the v2 kernel for a stencil: returns variant id or 0
// stem: (1,7,7) over 3 channels -> conv_wgrad_stem_kernel; fills the v2 plan fields
int pick_stem(const coclr_conv_desc* d, WPlan* w) {
  w->pw = 0;
  if (!(d->kt == 1 && d->kh == 7 && d->kw == 7 && d->Cin == 3)) return 0;
  ConvPlan& p = w->p2;
  conv_normalise(d, &p);
  conv_pick_box(&p, 7, 1, 7, 7);
  if (p.lTW < 1) return 0;
  // LDS row pitch of the window == 7 (mod 32): the (kh,kw) offsets of 32 consecutive columns
  // then hit distinct banks (the extra columns are loaded like any other, never read)
  while ((p.WW & 31) != 7) ++p.WW;
  p.plane1 = p.WT * p.WH * p.WW;
  p.plane = p.plane1 << p.lTN;
  if (cdiv(p.plane, 64) > 20) return 0;
  w->planeP2 = p.plane | 1;
  const size_t stage = ((size_t)64 * 129 + (size_t)3 * w->planeP2) * sizeof(float);
  w->lds2 = 2 * stage;
  if (w->lds2 > 160 * 1024) return 0;
  const double lim = 2147483648.0;
  if (((double)(1 << p.lTN) * d->x_nstride + (double)p.Cin * p.Ti * p.Hi * p.Wi) * 4.0 >= lim) return 0;
  if (((double)(1 << p.lTN) * d->y_nstride + (double)p.Cout * p.To * p.Ho * p.Wo) * 4.0 >= lim) return 0;
  w->mt2 = cdiv(p.Cout, 64);
  w->ct2 = 1;
  int S = 256 / w->mt2;                 // one 512-thread workgroup per CU
  if (S > p.ntiles / 2) S = p.ntiles / 2;
  if (S < 1) S = 1;
  w->S2 = S;
  return 9;
}

int pick_v2(const coclr_conv_desc* d, WPlan* w) {
  const int kt = d->kt, kh = d->kh, kw = d->kw;
  int MB = 1, NB = 1, id = 0;
  w->pw = 0;
  if (kt == 1 && kh == 3 && kw == 3) id = 1;
  else if (kt == 3 && kh == 1 && kw == 1) id = 2;
  else if (kt == 7 && kh == 1 && kw == 1) id = 3;
  else if (kt == 1 && kh == 1 && kw == 1) {
    if (d->st != 1 || d->sh != 1 || d->sw != 1) return 0;
    if (d->Cin > 64 && d->Cout > 64) { id = 4; MB = NB = 2; } else id = 5;
  } else return 0;
  // narrow layers: a 64co x 64ci workgroup tile would be mostly padding; the (ci,tap)-lane
  // kernel packs them better
  if (id <= 3 && (d->Cin < 48 || d->Cout < 48)) return 0;
  if (d->Cin < 8) return 0;
  ConvPlan& p = w->p2;
  conv_normalise(d, &p);
  if (id == 2 && d->algo >= 1 && d->st == 1 && d->pt == 1 && d->Ti == d->To && d->Cin >= 48 &&
      d->Cout >= 48) {
    // Winograd F(2,3) along T (the layer's forward / data gradient use it: desc.algo = 1): plan the
    // (4,1,1) / stride-2 view over frame PAIRS, 32 pairs per box
    ConvPlan q = p;
    q.To = (p.To + 1) / 2;
    q.st = 2;
    conv_pick_box(&q, 5, 4, 1, 1);
    const int planeP = q.plane | 1;
    const size_t stage = ((size_t)64 * 65 + (size_t)64 * planeP) * sizeof(float);
    const double lim = 2147483648.0;
    if (q.lTW >= 1 && cdiv(q.plane, 64) <= 2 && 2 * stage <= 160 * 1024 &&
        ((double)(1 << q.lTN) * d->x_nstride + (double)q.Cin * q.Ti * q.Hi * q.Wi) * 4.0 < lim &&
        ((double)(1 << q.lTN) * d->y_nstride + (double)q.Cout * p.To * q.Ho * q.Wo) * 4.0 < lim) {
      p = q;
      w->planeP2 = planeP;
      w->lds2 = 2 * stage;
      w->mt2 = cdiv(p.Cout, 64);
      w->ct2 = cdiv(p.Cin, 64);
      int S = 512 / (w->mt2 * w->ct2);
      if (S > p.ntiles / 4) S = p.ntiles / 4;
      if (S < 1) S = 1;
      w->S2 = xcd_round(S);
      return 6;
    }
  }
  // COCLR_WGRAD_WINO2: 0 never, 1 (default) the layers whose forward is Winograd (desc.algo = 1: maps of 16x16
  // and up), 2 every wide (1,3,3) stride-1 'same' layer on an even map
  static const int wino2_mode = getenv("COCLR_WGRAD_WINO2") ? atoi(getenv("COCLR_WGRAD_WINO2")) : 1;
  if (id == 1 && (d->algo == 1 || wino2_mode >= 2) && d->st == 1 && d->sh == 1 && d->sw == 1 && d->ph == 1 &&
      d->pw == 1 && d->pt == 0 && (d->Ho % 2) == 0 && (d->Wo % 2) == 0 && d->Hi == d->Ho && d->Wi == d->Wo) {
    // F(2x2,3x3) domain: 32 co x 64 ci per workgroup, boxes of 64 positions = 16 blocks whose pairs are
    // neighbours along w
    const bool off = wino2_mode <= 0;
    ConvPlan q = p;
    conv_pick_box(&q, 6, 1, 3, 3);
    if (q.lTW == 5 && q.lTH == 1 && q.lTT == 0 && q.lTN == 0 && q.Ho >= 4) {
      // 32 x 2 boxes stage a 34 x 4 window (three LDS-DMA pieces per channel); 16 x 4 boxes an 18 x 6 one
      // (two pieces): with 16 MFMAs per block pair instead of 36 the loader waves are what paces this
      // kernel, and a third fewer window pieces is a third less of their work
      q.lTW = 4; q.lTH = 2;
      q.nbw = cdiv(q.Wo, 16); q.nbh = cdiv(q.Ho, 4);
      q.ntiles = q.nbw * q.nbh * q.nbt * q.nbn;
      q.nboxes = q.ntiles;
      q.WH = 3 + 3; q.WW = 15 + 3;
      q.plane1 = q.WT * q.WH * q.WW;
      q.plane = q.plane1;
    }
    const int planeP = q.plane | 1;
    const size_t stage = ((size_t)32 * 65 + (size_t)64 * planeP) * sizeof(float);
    const double lim = 2147483648.0;
    if (!off && q.lTW >= 2 && q.lTH >= 1 && cdiv(q.plane, 64) <= 3 && 2 * stage <= 160 * 1024 &&
        ((double)(1 << q.lTN) * d->x_nstride + (double)q.Cin * q.Ti * q.Hi * q.Wi) * 4.0 < lim &&
        ((double)(1 << q.lTN) * d->y_nstride + (double)q.Cout * q.To * q.Ho * q.Wo) * 4.0 < lim) {
      p = q;
      w->planeP2 = planeP;
      w->lds2 = 2 * stage;
      w->mt2 = cdiv(p.Cout, 32);
      w->ct2 = cdiv(p.Cin, 64);
      int S = 512 / (w->mt2 * w->ct2);
      if (S > p.ntiles / 4) S = p.ntiles / 4;
      if (S < 1) S = 1;
      w->S2 = xcd_round(S);
      return 7;
    }
  }
  conv_pick_box(&p, 6, kt, kh, kw);
  if ((id == 4 || id == 5) && p.Ti == 1 && p.Hi == 1 && p.lTW == 6 && p.lTN == 0 &&
      (p.Wi % 64) == 0 && (d->x_nstride % 4) == 0 && (d->y_nstride % 4) == 0) {
    // contiguous 64-position boxes, everything 16-byte aligned: the 16-byte-DMA GEMM kernel
    // (pointer alignment is checked at launch)
    w->pw = 1;
  }
  if (p.lTW < 1) return 0;
  const int pch = cdiv(p.plane, 64);
  const int maxpch = id == 1 ? 3 : (id == 3 ? 4 : 2);
  if (pch > maxpch) return 0;
  w->planeP2 = p.plane | 1;
  const size_t stage = ((size_t)64 * MB * 65 + (size_t)64 * NB * w->planeP2) * sizeof(float);
  w->lds2 = 2 * stage;
  if (w->lds2 > 160 * 1024) return 0;
  // every byte offset formed inside a box stays below the 2 GiB descriptor range
  const double lim = 2147483648.0;
  if (((double)(1 << p.lTN) * d->x_nstride + (double)p.Cin * p.Ti * p.Hi * p.Wi) * 4.0 >= lim) return 0;
  if (((double)(1 << p.lTN) * d->y_nstride + (double)p.Cout * p.To * p.Ho * p.Wo) * 4.0 >= lim) return 0;
  w->mt2 = cdiv(p.Cout, 64 * MB);
  w->ct2 = cdiv(p.Cin, 64 * NB);
  // one 512-thread workgroup per CU; two rounds of workgroups, >= 4 boxes each
  int S = 512 / (w->mt2 * w->ct2);
  if (S > p.ntiles / 4) S = p.ntiles / 4;
  if (S < 1) S = 1;
  w->S2 = xcd_round(S);
  return id;
}

// Policy (measured per launch on one MI355X, tools/wgrad_order_ab.py, profiles/r04_wgrad_order.txt; traffic
// in profiles/r04_pmc_wgrad.txt): tile-fastest XCD-aware ids when ONE SAMPLE of the input no longer fits an
// XCD's 4 MiB L2.  Above that size, which XCD streams which boxes decides whether the halo rows of a
// (1,3,3) window and the other half of a 128-byte line of a 16-float temporal row are fetched once or once
// per XCD (Conv_2c.conv2 0.898 -> 0.863 ms, Conv_1a.conv2 1.051 -> 0.992); below it every L2 ends up
// holding the whole sample either way and the launch grid's own order dispatches a few percent better
// (128->128 (3,1,1) on 8x8x8: 0.023 vs 0.027 ms).
inline int wgrad_tile_fastest(const coclr_conv_desc* d) {
  const int forced = wgrad_order_env();
  if (forced >= 0) return forced;
  return (double)d->Cin * d->Ti * d->Hi * d->Wi * 4.0 >= 4.0 * 1024 * 1024;
}

int plan_wgrad(const coclr_conv_desc* d, WPlan* w) {
  if (!d || d->N <= 0 || d->Cin <= 0 || d->Cout <= 0) return COCLR_EINVAL;
  if (d->dt != 1 || d->dh != 1 || d->dw != 1) return COCLR_EINVAL;
  w->v2 = pick_stem(d, w);
  if (w->v2) return 0;
  w->v2 = pick_v2(d, w);
  if (w->v2) return 0;
  ConvPlan& p = w->p;
  conv_normalise(d, &p);
  const int taps = d->kt * d->kh * d->kw;
  conv_pick_box(&p, 7, d->kt, d->kh, d->kw);
  const int J = p.Cin * taps;
  if (p.WT > 255 || p.WH > 255 || p.WW > 255 || (1 << p.lTN) > 255) return COCLR_EINVAL;
  w->pch = cdiv(p.plane, 64);
  w->planeP = w->pch * 64 + 1;
  w->BJ = taps == 1 ? 64 : 128;
  w->nci_max = taps == 1 ? 64 : (w->BJ - 1) / taps + 2;
  if (w->nci_max > p.Cin) w->nci_max = p.Cin;
  // variant = PCH bucket
  if (w->pch <= 2) w->variant = 0;
  else if (w->pch <= 4) w->variant = 1;
  else if (w->pch <= 8) w->variant = 2;
  else if (w->pch <= 20) w->variant = 3;
  else return COCLR_EINVAL;
  const size_t lds = ((size_t)64 * 129 + 128 + (size_t)w->nci_max * w->planeP) * sizeof(float);
  if (lds > 160 * 1024) return COCLR_EINVAL;
  w->jtiles = cdiv(J, w->BJ);
  w->mtiles = cdiv(p.Cout, 64);
  // split-K: enough workgroups to fill 256 CUs x ~3, but at least 4 boxes per workgroup so
  // the partial-sum traffic stays small next to the streamed operands
  int S = 768 / (w->jtiles * w->mtiles);
  if (S > p.ntiles / 4) S = p.ntiles / 4;
  if (S < 1) S = 1;
  w->S = xcd_round(S);
  return 0;
}

template <int BJ, int PCH>
int launch_wgrad(WgradArgs& a, const WPlan& w, hipStream_t stream) {
  const size_t lds = ((size_t)64 * 129 + 128 + (size_t)w.nci_max * a.planeP) * sizeof(float);
  auto kern = conv_wgrad_kernel<BJ, PCH>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3(a.S, a.jtiles, a.mtiles), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int KT, int KH, int KW, int MB, int NB, int PCH, bool WINO = false, int NL = 4, bool WHW = false>
int launch_wgrad2(const Wgrad2Args& a, const WPlan& w, hipStream_t stream) {
  auto kern = conv_wgrad2_kernel<KT, KH, KW, MB, NB, PCH, WINO, NL, WHW>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3(w.S2, w.ct2, w.mt2), dim3(256 + 64 * NL), w.lds2, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int coclr_conv3d_wgrad_workspace(const coclr_conv_desc* d, int64_t* elems) {
  WPlan w;
  int rc = plan_wgrad(d, &w);
  if (rc) return rc;
  // split slices: the stem kernel writes four per split (one per matrix wave), the F(2x2,3x3) form two
  *elems = (int64_t)(w.v2 ? w.S2 * (w.v2 == 9 ? 4 : (w.v2 == 7 ? 2 : 1)) : w.S) * d->Cout * d->Cin * d->kt *
           d->kh * d->kw;
  return 0;
}

namespace {
// BatchNorm backward apply pass folded into the weight gradient (stem kernel, BNA form)
struct WgradBn {
  const float* y;
  const float* coef;
  long y_nstride;
  int relu;
};
// LDS of the stem kernel's two stages in the BNA form
inline size_t stem_bna_lds(const WPlan& w) {
  return 2 * ((size_t)2 * 64 * 129 + 128 + (size_t)3 * w.planeP2) * sizeof(float);
}
int wgrad_run(const coclr_conv_desc* d, const float* x, const float* dy, const WgradBn* bn,
              float* const* dw_list, const int32_t* row_end, int nseg, float* workspace,
              int64_t w_co_stride, int64_t w_ci_stride, int tap_base, int accumulate, void* stream_);
}  // namespace

extern "C" int coclr_conv3d_wgrad(const coclr_conv_desc* d, const float* x, const float* dy,
                                  float* dw, float* workspace, int64_t w_co_stride,
                                  int64_t w_ci_stride, int tap_base, int accumulate,
                                  void* stream_) {
  float* one[1] = {dw};
  const int32_t end[1] = {d ? d->Cout : 0};
  return coclr_conv3d_wgrad_multi(d, x, dy, one, end, 1, workspace, w_co_stride, w_ci_stride,
                                  tap_base, accumulate, stream_);
}

extern "C" int coclr_conv3d_wgrad_multi(const coclr_conv_desc* d, const float* x, const float* dy,
                                        float* const* dw_list, const int32_t* row_end, int nseg,
                                        float* workspace, int64_t w_co_stride,
                                        int64_t w_ci_stride, int tap_base, int accumulate,
                                        void* stream_) {
  return wgrad_run(d, x, dy, nullptr, dw_list, row_end, nseg, workspace, w_co_stride, w_ci_stride, tap_base,
                   accumulate, stream_);
}

extern "C" int coclr_conv3d_wgrad_bn_ok(const coclr_conv_desc* d, int* ok) {
  if (!d || !ok) return COCLR_EINVAL;
  WPlan w;
  int rc = plan_wgrad(d, &w);
  if (rc) return rc;
  *ok = (w.v2 == 9 && stem_bna_lds(w) <= (size_t)160 * 1024) ? 1 : 0;
  return 0;
}

extern "C" int coclr_conv3d_wgrad_bn(const coclr_conv_desc* d, const float* x, const float* dz, const float* y,
                                     int64_t y_nstride, const float* coef, int relu, float* dw,
                                     float* workspace, int64_t w_co_stride, int64_t w_ci_stride, int tap_base,
                                     int accumulate, void* stream_) {
  if (!d || !y || !coef) return COCLR_EINVAL;
  float* one[1] = {dw};
  const int32_t end[1] = {d->Cout};
  const WgradBn bn = {y, coef, (long)y_nstride, relu ? 1 : 0};
  return wgrad_run(d, x, dz, &bn, one, end, 1, workspace, w_co_stride, w_ci_stride, tap_base, accumulate,
                   stream_);
}

namespace {
int wgrad_run(const coclr_conv_desc* d, const float* x, const float* dy, const WgradBn* bn,
              float* const* dw_list, const int32_t* row_end, int nseg, float* workspace,
              int64_t w_co_stride, int64_t w_ci_stride, int tap_base, int accumulate, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d || !dw_list || !row_end || nseg < 1 || nseg > 4) return COCLR_EINVAL;
  WgradDst dst;
  dst.n = nseg;
  for (int i = 0; i < 4; ++i) {
    dst.p[i] = i < nseg ? dw_list[i] : nullptr;
    dst.end[i] = i < nseg ? row_end[i] : 0;
    if (i < nseg && (!dw_list[i] || row_end[i] <= (i ? row_end[i - 1] : 0))) return COCLR_EINVAL;
  }
  if (row_end[nseg - 1] != d->Cout) return COCLR_EINVAL;
  WPlan w;
  int rc = plan_wgrad(d, &w);
  if (rc) return rc;
  if (bn && !(w.v2 == 9 && stem_bna_lds(w) <= (size_t)160 * 1024)) return COCLR_EINVAL;   // see _bn_ok
  const int taps = d->kt * d->kh * d->kw;
  int S_used;
  if (w.v2) {
    const ConvPlan& p = w.p2;
    Wgrad2Args a;
    a.x = x; a.dy = dy; a.part = workspace;
    a.x_nstride = d->x_nstride; a.dy_nstride = d->y_nstride;
    a.x_cstride = p.Ti * p.Hi * p.Wi; a.dy_cstride = p.To * p.Ho * p.Wo;
    a.N = p.N; a.Cin = p.Cin; a.Cout = p.Cout; a.J = p.Cin * taps;
    a.Ti = p.Ti; a.Hi = p.Hi; a.Wi = p.Wi; a.To = p.To; a.Ho = p.Ho; a.Wo = p.Wo;
    a.st = p.st; a.sh = p.sh; a.sw = p.sw; a.pt = p.pt; a.ph = p.ph; a.pw = p.pw;
    a.lTW = p.lTW; a.lTH = p.lTH; a.lTT = p.lTT; a.lTN = p.lTN;
    a.nbw = p.nbw; a.nbh = p.nbh; a.nbt = p.nbt; a.nbn = p.nbn;
    a.WT = p.WT; a.WH = p.WH; a.WW = p.WW; a.plane1 = p.plane1; a.plane = p.plane;
    a.planeP = w.planeP2;
    a.inv_plane1 = 1.0f / (float)p.plane1;
    a.inv_hw = 1.0f / (float)(p.WH * p.WW);
    a.inv_ww = 1.0f / (float)p.WW;
    a.ntiles = p.ntiles; a.S = w.S2;
    a.To_full = d->To;
    a.xcd = wgrad_tile_fastest(d);
    a.bn_y = bn ? bn->y : nullptr; a.bn_coef = bn ? bn->coef : nullptr;
    a.bn_y_nstride = bn ? bn->y_nstride : 0; a.bn_relu = bn ? bn->relu : 0;
    if (w.v2 == 6) a.dy_cstride = d->To * p.Ho * p.Wo;     // p.To counts pairs there
    const int pch = cdiv(p.plane, 64);
    const bool pw = w.pw && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dy % 16) == 0;
    if (pw) {
      const bool big = w.v2 == 4;
      const size_t lds = (size_t)2 * (big ? 256 : 128) * 64 * sizeof(float);
      if (big) {
        auto kern = conv_wgrad_pw_kernel<2, 2>;
        static std::atomic<uint64_t> attr_done{0};
        COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
        hipLaunchKernelGGL(kern, dim3(w.S2, w.ct2, w.mt2), dim3(512), lds, stream, a);
      } else {
        auto kern = conv_wgrad_pw_kernel<1, 1>;
        static std::atomic<uint64_t> attr_done{0};
        COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
        hipLaunchKernelGGL(kern, dim3(w.S2, w.ct2, w.mt2), dim3(512), lds, stream, a);
      }
      COCLR_LAUNCH_CHECK();
      rc = 0;
    } else
    switch (w.v2) {
      case 1: rc = pch <= 2 ? launch_wgrad2<1, 3, 3, 1, 1, 2>(a, w, stream)
                            : launch_wgrad2<1, 3, 3, 1, 1, 3>(a, w, stream); break;
      // the temporal forms are loader-bound with four loader waves (3 or 4 MFMAs per four LDS-DMA
      // pieces): EIGHT loader waves beside the four matrix waves (768 threads; 48-64 accumulator
      // registers leave room for three waves per SIMD): direct 1.20 -> 1.05 ms, Winograd 1.07 ->
      // 0.90 ms on Conv_2c.conv2.  (7,1,1) (112 accumulators) and (1,3,3) (144) keep four.
      case 2: rc = launch_wgrad2<3, 1, 1, 1, 1, 2, false, 8>(a, w, stream); break;
      case 3: rc = launch_wgrad2<7, 1, 1, 1, 1, 4>(a, w, stream); break;
      case 4: rc = launch_wgrad2<1, 1, 1, 2, 2, 2>(a, w, stream); break;
      case 5: rc = launch_wgrad2<1, 1, 1, 1, 1, 2>(a, w, stream); break;
      case 6: rc = launch_wgrad2<4, 1, 1, 1, 1, 2, true, 8>(a, w, stream); break;
      case 7: rc = pch <= 2 ? launch_wgrad2<1, 3, 3, 1, 1, 2, false, 4, true>(a, w, stream)
                            : launch_wgrad2<1, 3, 3, 1, 1, 3, false, 4, true>(a, w, stream); break;
      case 9: {
        if (bn) {
          auto kern = conv_wgrad_stem_kernel<7, 7, 3, 20, true>;
          static std::atomic<uint64_t> attr_done{0};
          COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
          hipLaunchKernelGGL(kern, dim3(w.S2, 1, w.mt2), dim3(512), stem_bna_lds(w), stream, a);
        } else {
          auto kern = conv_wgrad_stem_kernel<7, 7, 3, 20>;
          static std::atomic<uint64_t> attr_done{0};
          COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
          hipLaunchKernelGGL(kern, dim3(w.S2, 1, w.mt2), dim3(512), w.lds2, stream, a);
        }
        COCLR_LAUNCH_CHECK();
        rc = 0;
        break;
      }
      default: rc = COCLR_EINVAL;
    }
    if (rc) return rc;
    S_used = w.S2 * (w.v2 == 9 ? 4 : (w.v2 == 7 ? 2 : 1));
  } else {
    const ConvPlan& p = w.p;
    WgradArgs a;
    a.x = x; a.dy = dy; a.part = workspace;
    a.x_nstride = d->x_nstride; a.dy_nstride = d->y_nstride;
    a.x_cstride = p.Ti * p.Hi * p.Wi; a.dy_cstride = p.To * p.Ho * p.Wo;
    a.N = p.N; a.Cin = p.Cin; a.Cout = p.Cout;
    a.taps = taps; a.KH = d->kh; a.KW = d->kw;
    a.J = p.Cin * a.taps;
    a.Ti = p.Ti; a.Hi = p.Hi; a.Wi = p.Wi; a.To = p.To; a.Ho = p.Ho; a.Wo = p.Wo;
    a.st = p.st; a.sh = p.sh; a.sw = p.sw; a.pt = p.pt; a.ph = p.ph; a.pw = p.pw;
    a.lTW = p.lTW; a.lTH = p.lTH; a.lTT = p.lTT; a.lTN = p.lTN;
    a.nbw = p.nbw; a.nbh = p.nbh; a.nbt = p.nbt; a.nbn = p.nbn;
    a.WT = p.WT; a.WH = p.WH; a.WW = p.WW; a.plane1 = p.plane1; a.plane = p.plane;
    a.planeP = w.planeP; a.pch = w.pch;
    a.ntiles = p.ntiles; a.S = w.S; a.jtiles = w.jtiles; a.mtiles = w.mtiles;
    const bool pw = w.BJ == 64;
    switch (w.variant) {
      case 0: rc = pw ? launch_wgrad<64, 2>(a, w, stream) : launch_wgrad<128, 2>(a, w, stream); break;
      case 1: rc = pw ? launch_wgrad<64, 4>(a, w, stream) : launch_wgrad<128, 4>(a, w, stream); break;
      case 2: rc = pw ? launch_wgrad<64, 8>(a, w, stream) : launch_wgrad<128, 8>(a, w, stream); break;
      case 3: rc = pw ? launch_wgrad<64, 20>(a, w, stream) : launch_wgrad<128, 20>(a, w, stream); break;
      default: rc = COCLR_EINVAL;
    }
    if (rc) return rc;
    S_used = w.S;
  }
  const int J = d->Cin * taps;
  const long CJ = (long)d->Cout * J;
  int blocks = cdiv(CJ, 64);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, workspace, dst, CJ,
                     S_used, J, taps, (long)w_co_stride, (long)w_ci_stride, tap_base, accumulate);
  COCLR_LAUNCH_CHECK();
  return 0;
}
}  // namespace
