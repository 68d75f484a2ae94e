// Implicit-GEMM 3D convolution for gfx950 on NCDHW fp32 tensors.
//
//   Y[n][co][o] = sum_{ci,tap} W[co][ci][tap] * V[n][ci][o*stride - pad + tap]
//
// where V is the input, optionally zero-upsampled ("input dilation") so the
// same kernel serves as the data-gradient of a strided convolution, and the
// output positions may be a strided sub-lattice of the destination tensor
// (phase-decomposed data gradient).  This one kernel family replaces every
// ATen/cuDNN conv3d the reference issues from backbone/s3dg.py:11-13,39-42 and
// backbone/resnet_2d3d.py:53-59,138 (forward), and the dgrad half of their
// autograd backward.
//
// Mapping to the hardware:
//   * GEMM view: M = Cout, N = output positions (a power-of-two 4-D "box"
//     n x t x h x w owned by one workgroup), K = Cin x taps, in chunks of CC
//     input channels.
//   * Both operands of a chunk are brought in by the LDS-DMA engine
//     (buffer_load_dword[x4] ... lds): no staging registers, out-of-range
//     lanes (zero padding, box overhang, dilation holes) are dropped by the
//     buffer bounds check and land as 0.0.  Two LDS stages: chunk i+1 streams
//     in while chunk i is multiplied; one barrier per chunk.
//   * The input stencil window of the box sits in LDS as [c][window]
//     (positions contiguous); every tap reads it at a shifted offset, so
//     HBM/L2 sees each input element ~once per box instead of once per tap.
//   * Weights arrive pre-packed as [tap][CinP][CoutP] (Cout contiguous, padded
//     to x128) and are staged as [tap][c][BM]; both MFMA operand reads are
//     stride-1 across the 32 lanes of a half-wave -> conflict-free ds_read_b32.
//   * Math: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain).  Lanes 0-31 carry
//     k = even channel of the chunk, lanes 32-63 the odd one, same tap.
//     Operands of step i+1 are fetched while the MFMAs of step i issue.
//   * 256 threads = 4 waves as 2(M) x 2(N).
//   * Epilogue: buffer stores with scalar row offsets (no per-element address
//     math or exec masking), optional bias / per-channel affine / ReLU /
//     accumulate, plus per-workgroup partial sums (sum, sum of squares) per
//     output channel for train-mode BatchNorm, reduced with DPP row operations
//     and written without atomics as [2][Cout][ntiles].
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_geom.h"

namespace {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr unsigned OOB = 0x80000000u;        // byte offset no buffer below reaches
constexpr unsigned BUF_RANGE = 0x80000000u;  // num_records of every descriptor

struct ConvArgs {
  const float* x;
  const float* w;        // packed [taps][CinP][CoutP]
  float* y;
  float* stats;          // [2][Cout][ntiles] or nullptr
  const float* bias;     // [Cout] or nullptr
  const float* ep_scale; // [Cout] or nullptr
  const float* ep_shift; // [Cout] or nullptr
  const int64_t* n_index;// optional gather of input samples
  // data gradient that lands in the dz of a BatchNorm(+ReLU) unit: the unit's conv output (laid out like
  // y) and coefficients; the statistics slots then receive (sum g, sum g*xhat) of that unit's backward
  const float* bwd_y;
  const float* bwd_scale;
  const float* bwd_shift;
  const float* bwd_mean;
  const float* bwd_invstd;
  int bwd_relu;
  long x_nstride, y_nstride;
  int x_cstride, y_cstride;
  int N, Cin, Cout, CinP, CoutP;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw, dt, dh, dw;
  int yHf, yWf;                       // full H, W extent of the destination tensor
  int yst, ysh, ysw, yot, yoh, yow;   // destination position = o * ys + yo
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int WT, WH, WW, plane1, plane, planeS;
  float inv_plane1, inv_hw, inv_ww;
  int mtiles, ntiles, nchunks;
  int relu, accumulate;
  int xcd;                // remap workgroup ids so that an XCD owns contiguous tiles
  int To_full;            // F(4,3) / F(2,4) temporal kernels: output frames of the launch (a.To counts groups)
  // the input is the raw output of a BatchNorm(+ReLU) unit: x' = max?(x * in_scale[ci] + in_shift[ci], 0) is
  // applied while the kernel reads (conv_poly7_body INAFF), zero padding stays zero
  const float* in_scale;
  const float* in_shift;
  int in_relu;
};

// sum over each 16-lane row (result in every lane of the row)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_amdgcn_update_dpp(0.f, v, 0xB1, 0xf, 0xf, true);    // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x4E, 0xf, 0xf, true);    // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x141, 0xf, 0xf, true);   // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0.f, v, 0x140, 0xf, 0xf, true);   // row_mirror
  return v;
}

// exact floor(e / d) for 0 <= e < 2^20 given inv = 1.0f / d
__device__ __forceinline__ int fdiv(int e, float inv) { return (int)(((float)e + 0.5f) * inv); }

// LDS-DMA of 4 / 16 bytes per lane as plain __device__ functions: inside a template (the kernels' lambdas
// are implicitly __host__ __device__ and re-checked at instantiation) the host pass, which has no
// gfx950 features, rejects the 16-byte size.
__device__ __forceinline__ void dma4_to_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned voff,
                                            unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_dst), 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void dma16_to_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned voff,
                                             unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(lds_dst), 16, voff, soff, 0, 0);
}

// XG: spatial stencils (kw = 3, stride 1, pad 1, no dilation) stage their window with 16-BYTE LDS-DMA:
// rows are widened to whole 16-byte granules of the input row (global columns [ow0 - 4, ow0 + TW + 4),
// so window column 0 sits at LDS column 3), a lane of a DMA piece moves one granule, and a 10x10 window
// takes ONE piece per channel instead of two scattered 4-byte ones.  a.WW / a.plane1 are then the padded
// row / sample extents in floats, a.plane the number of granules, a.inv_* reciprocals in granules.
template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4 = false, bool XG = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, int bid, const int nblocks) {
  constexpr int TAPS = KT * KH * KW;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BN / (WN * 32);
  static_assert(CC % 4 == 0, "chunk: every wave stages CC/4 channels");
  static_assert(MF >= 1 && NF >= 1, "tile shape");
  constexpr int RPP = 256 / BM;                 // weight rows per 1 KiB DMA piece
  constexpr int WPIECES = TAPS * CC / RPP;      // pieces per chunk
  static_assert(CC % RPP == 0, "a DMA piece never straddles two taps");
  constexpr int W_FLOATS = TAPS * CC * BM;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- which tile --------------------------------------------------------
  // XCD-aware: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, each with its
  // own L2.  Logical tile ids are remapped so that XCD x owns a CONTIGUOUS run of them: the cout tiles
  // of one box of positions (which all read the same input window) then share an L2 instead of
  // fetching the window once per XCD.
  if (a.xcd) {
    const int per = nblocks >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
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

  // descriptors: x relative to the box's first sample (or to the tensor when samples are
  // gathered), packed weights, y relative to the box's first sample
  const bool gather = a.n_index != nullptr;
  const float* xbase = gather ? a.x : a.x + (long)n0 * a.x_nstride;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);
  float* ybase = a.y + (long)n0 * a.y_nstride;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);

  // ---- per-lane byte offsets of the window elements this lane's DMA pieces fetch ----
  unsigned goff[PCH];
  {
    const int rowlen = XG ? a.WW >> 2 : a.WW;          // XG: in granules
    const int hw = a.WH * rowlen;
    const int p1 = XG ? a.plane1 >> 2 : a.plane1;
    const bool dil = !XG && (a.dt | a.dh | a.dw) != 1;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned off = OOB;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * p1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * rowlen;
        const int n = n0 + wn_;
        // XG: granule g of a row starts at column ow0 - 4 + 4g: wholly inside or outside (Wi % 4 == 0)
        int it = vt0 + wt, ih = vh0 + wh, iw = XG ? vw0 - 3 + 4 * ww : vw0 + ww;
        bool ok = n < a.N && it >= 0 && ih >= 0 && iw >= 0;
        if (dil && ok) {
          const int vt = it, vh = ih, vw = iw;
          it = vt / a.dt; ih = vh / a.dh; iw = vw / a.dw;
          ok = it * a.dt == vt && ih * a.dh == vh && iw * a.dw == vw;
        }
        if (ok && it < a.Ti && ih < a.Hi && iw < a.Wi) {
          const long ns = gather ? (long)a.n_index[n] : (long)wn_;
          off = (unsigned)((ns * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4);
        }
      }
      goff[j] = off;
    }
  }
  // weights: lane's row inside a piece and column
  const unsigned wvoff = (unsigned)((((lane * 4) / BM) * a.CoutP + (lane * 4) % BM) * 4);
  // XV4: stencils that do not reach along the flattened (H,W) axis (pointwise and temporal
  // ones) have windows without a W halo; with everything 16-byte aligned the LDS image
  // [c][plane] is filled by 16-byte DMA in 1 KiB pieces that run across channel rows: piece j,
  // lane L holds floats 256j + 4L .. +3 of the chunk image.  Wave w owns pieces w, w+4, ...
  constexpr int PV = XV4 ? (CC * PCH * 64 / 256 + 3) / 4 : 1;
  unsigned xvoff[PV];
  int xvc[PV];
  if (XV4) {
#pragma unroll
    for (int jj = 0; jj < PV; ++jj) {
      const int flat = (wave + 4 * jj) * 256 + lane * 4;
      const int c = flat / plane, e = flat - c * plane;
      unsigned off = OOB;
      if (c < CC) {
        const int wn_ = fdiv(e, a.inv_plane1);
        const int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_ww);              // WH == 1 here
        const int ww = q - wt * a.WW;
        const int n = n0 + wn_, it = vt0 + wt, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && iw < a.Wi) {
          const long ns = gather ? (long)a.n_index[n] : (long)wn_;
          off = (unsigned)((ns * a.x_nstride + (long)it * a.Wi + iw + (long)c * a.x_cstride) * 4);
        }
      }
      xvoff[jj] = off;
      xvc[jj] = c;
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
    lanebase[nf] = W_FLOATS + tn * a.plane1 + ((tt * a.st) * a.WH + th * a.sh) * a.WW +
                   tw * a.sw + half * planeS + (XG ? 3 : 0);
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  f32x16 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mf][nf][i] = 0.f;

  // ---- DMA of one chunk into stage `sbase` ------------------------------------
  auto stage = [&](int cin0, float* sbase) {
    // weights: pieces wave, wave+4, ...
    for (int p = wave; p < WPIECES; p += 4) {
      const int row0 = p * RPP;
      const int tap = row0 / CC, c0 = row0 % CC;
      const unsigned soff = (unsigned)((((long)tap * a.CinP + cin0 + c0) * a.CoutP + cout0) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sbase + p * 256), 16, wvoff, soff, 0, 0);
    }
    // input window: channels wave, wave+4, ...
    float* xs = sbase + W_FLOATS;
    if (XV4) {
      const unsigned soff = (unsigned)cin0 * (unsigned)a.x_cstride * 4u;
#pragma unroll
      for (int jj = 0; jj < PV; ++jj) {
        const int j = wave + 4 * jj;
        if (j * 256 < CC * plane && xvc[jj] < CC) {   // exec-masked: lanes past the image write nothing
          const unsigned vo = cin0 + xvc[jj] < a.Cin ? xvoff[jj] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + j * 256), 16, vo, soff, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) {
            if (XG) dma16_to_lds(rx, xs + c * planeS + j * 256, goff[j], soff);
            else dma4_to_lds(rx, xs + c * planeS + j * 64, goff[j], soff);
          }
      } else {
        // channel past Cin: its packed weights are zero, keep the operand finite
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) {
            if (XG)
              *reinterpret_cast<float4*>(&xs[c * planeS + j * 256 + lane * 4]) =
                  make_float4(0.f, 0.f, 0.f, 0.f);
            else
              xs[c * planeS + j * 64 + lane] = 0.f;
          }
      }
    }
  };

  // ---- main loop ---------------------------------------------------------------
  const int nchunks = a.nchunks;
  stage(0, smem);
  for (int ch = 0; ch < nchunks; ++ch) {
    float* cur = smem + (ch & 1) * stage_floats;
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): this wave's DMA of chunk ch has landed
    __syncthreads();                      // ... everyone's has, and chunk ch-1 is fully consumed
    if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((ch + 1) & 1) * stage_floats);

    // Operands of MFMA step i+1 are fetched from LDS before the MFMAs of step i issue
    // (register double buffer), so the LDS latency sits under matrix-pipe time.
    constexpr int QS = CC / 2;
    constexpr int STEPS_T = KH * KW * QS;      // steps per temporal tap
    auto fetch = [&](int kt, int i, float (&av)[MF], float (&bv)[NF]) {
      const int rr = i / QS, q = i % QS;
      const int kh = rr / KW, kw = rr % KW;
      const int tap = (kt * KH + kh) * KW + kw;
      const int tapoff = (kt * a.WH + kh) * a.WW + kw;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) av[mf] = cur[abase + (tap * CC + 2 * q) * BM + mf * 32];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bv[nf] = cur[lanebase[nf] + tapoff + 2 * q * planeS];
    };
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      float av[2][MF], bv[2][NF];
      fetch(kt, 0, av[0], bv[0]);
#pragma unroll
      for (int i = 0; i < STEPS_T; ++i) {
        if (i + 1 < STEPS_T) fetch(kt, i + 1, av[(i + 1) & 1], bv[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                av[i & 1][mf], bv[i & 1][nf], acc[mf][nf], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------
  // destination byte offset of this lane's column (relative to ybase, channel row added
  // as a scalar), OOB when the position lies outside the tensor
  unsigned yvoff[NF];
  bool pvalid[NF];
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BN / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, ot = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    pvalid[nf] = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
    const long e = (long)tn * a.y_nstride +
                   ((long)(ot * a.yst + a.yot) * a.yHf + (oh * a.ysh + a.yoh)) * a.yWf +
                   (ow * a.ysw + a.yow);
    yvoff[nf] = pvalid[nf] ? (unsigned)(e * 4) + half_rows : OOB;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;  // [4 (wn, 16-lane row)][BM][2], reused after the main loop
  if (want_stats) __syncthreads();

  // FANCY: bias / affine / ReLU asked for (inference epilogue).  The plain form must not even CONTAIN the
  // per-row coefficient loads: a conditional global load in the store loop makes the compiler wait vmcnt(0)
  // in front of every row's stores, i.e. for all earlier stores to land -- sixteen memory round trips per tile.
  auto emit = [&](auto acc_tag, auto fancy_tag) {
    constexpr bool ACCUM = decltype(acc_tag)::value;
    constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      // accumulate: ALL old values of this row block first -- one round trip; loaded next to its store,
      // every element waited for its own (the compiler cannot move a load above the store before it):
      // the fused heads of Mixed_3c, whose data gradient adds to the pool branch's, 0.309 -> 0.2 ms
      float old[ACCUM ? 16 : 1][ACCUM ? NF : 1];
      if (ACCUM) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
          const bool cok = cout0 + rowu + 4 * half < a.Cout;
          const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            old[ACCUM ? i : 0][ACCUM ? nf : 0] = __uint_as_float(
                __builtin_amdgcn_raw_buffer_load_b32(ry, cok ? yvoff[nf] : OOB, soff, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);   // wave-uniform
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float s = 0.f, ss = 0.f;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const unsigned vo = cok ? yvoff[nf] : OOB;
          float v = acc[mf][nf][i];
          if (ACCUM) v += old[ACCUM ? i : 0][ACCUM ? nf : 0];
          const float vm = pvalid[nf] ? v : 0.f;
          s += vm; ss += vm * vm;
          if (FANCY) {
            v += bia;
            v = v * sc + sf;
            if (a.relu) v = fmaxf(v, 0.f);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, vo, soff, 0);
        }
        if (want_stats) {
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  // The destination is the dz of a BatchNorm(+ReLU) unit (a.bwd_y = that unit's conv output): store dz and
  // form the unit's backward sums -- g = dz where its ReLU passed, (sum g, sum g*xhat) -- from the values
  // in registers, in the arithmetic of bn_act_bwd_reduce_kernel (csrc/bn.hip), whose pass over dz and y
  // this replaces by one read of y here (aten::native_batch_norm_backward's reduction, threshold_backward).
  auto emit_bwd = [&]() {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.bwd_y + (long)n0 * a.y_nstride), 0, BUF_RANGE, 0x00020000);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      float yv[16][NF];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const bool cok = cout0 + rowu + 4 * half < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          yv[i][nf] = __uint_as_float(
              __builtin_amdgcn_raw_buffer_load_b32(rb, cok ? yvoff[nf] : OOB, soff, 0));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float bsc = 0.f, bsf = 0.f, mu = 0.f, is = 0.f;
        if (cok) { bsc = a.bwd_scale[co]; bsf = a.bwd_shift[co]; mu = a.bwd_mean[co]; is = a.bwd_invstd[co]; }
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float v = acc[mf][nf][i];
          const float u = yv[i][nf];
          const bool on = pvalid[nf] && (!a.bwd_relu || fmaf(u, bsc, bsf) > 0.f);
          const float g = on ? v : 0.f;
          s += g; ss += g * ((u - mu) * is);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, cok ? yvoff[nf] : OOB, soff, 0);
        }
        s = row16_sum(s);
        ss = row16_sum(ss);
        if ((lane & 15) == 0) {
          const int slot = wn * 2 + (l31 >> 4);
          red[(slot * BM + ml) * 2 + 0] = s;
          red[(slot * BM + ml) * 2 + 1] = ss;
        }
      }
    }
  };
  const bool fancy = a.bias || a.ep_scale || a.relu;
  if (a.bwd_y) emit_bwd();
  else if (fancy) { if (a.accumulate) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{}); }
  else if (a.accumulate) emit(std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{});

  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[(k * BM + tid) * 2];
          ss += red[(k * BM + tid) * 2 + 1];
        }
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}

template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4 = false, bool XG = false>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(const ConvArgs a) {
  conv_igemm_body<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>(a, (int)blockIdx.x, (int)gridDim.x);
}

// TWO problems of the same kernel variant in one launch: workgroups [0, nb0) run problem a0, the rest
// problem a1 (the two separable branches of an inception block, backbone/s3dg.py:100-118, share every
// stencil shape: on the 8x8x8 / 4x4x4 maps each of them alone is a 10-40 us launch that leaves most of
// the chip idle).  Each half keeps its own XCD-aware tile numbering.
template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4 = false, bool XG = false>
__global__ void __launch_bounds__(256)
conv_igemm_pair_kernel(const ConvArgs a0, const ConvArgs a1, const int nb0) {
  if ((int)blockIdx.x < nb0)
    conv_igemm_body<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>(a0, (int)blockIdx.x, nb0);
  else
    conv_igemm_body<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>(a1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0);
}

// ------------------------------------------------------------------------------------
// Temporal (3,1,1) stride-1 convolutions through Winograd F(2,3) along T.
//
// A pair of output frames (2p, 2p+1) needs input frames d0..d3 = 2p-1 .. 2p+2 and FOUR channel
// contractions instead of six:
//     m0 = G0 (d0 - d2)   m1 = G1 (d1 + d2)   m2 = G2 (d2 - d1)   m3 = G3 (d1 - d3)
//     y[2p] = m0 + m1 + m2          y[2p+1] = m1 - m2 - m3
// with G0 = w0, G1 = (w0+w1+w2)/2, G2 = (w0-w1+w2)/2, G3 = w2 ([Cout][Cin] matrices, prepared
// by pack_weights_kernel as a 4-"tap" operand).  Everything stays fp32; measured against float64
// the result is as accurate as the direct fp32 convolution (tools/winograd_probe.py: 5e-7).
//
// Seen from the implicit-GEMM kernel above this is a (4,1,1) stencil with temporal stride 2
// over "pair positions": same box / window / LDS-DMA staging / weight tile; what differs is that
// the B operand of virtual tap i is a difference or sum of two window rows (one VALU op), that a
// wave keeps four accumulators per 32x32 block, and that the epilogue emits two frames.
// 1.5x fewer MFMAs for the (3,1,1) layers (27 % of the S3D conv FLOPs), forward and dgrad.
template <int CC, int BM, int BNP, int PCH, bool XV4>
__device__ __forceinline__ void conv_wino_t_body(const ConvArgs& a, int bid, const int nblocks) {
  constexpr int TAPS = 4;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BNP / (WN * 32);
  constexpr int RPP = 256 / BM;
  constexpr int WPIECES = TAPS * CC / RPP;
  static_assert(CC % RPP == 0 && CC % 4 == 0, "chunk shape");
  constexpr int W_FLOATS = TAPS * CC * BM;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  if (a.xcd) {                        // XCD-aware tile ids (see conv_igemm_body)
    const int per = nblocks >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const int mt = bid % a.mtiles;
  const int ntile = bid / a.mtiles;
  int r = ntile;
  const int bw_ = r % a.nbw; r /= a.nbw;
  const int bh_ = r % a.nbh; r /= a.nbh;
  const int bt_ = r % a.nbt; r /= a.nbt;
  const int n0 = r << a.lTN;
  const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;   // ot0: PAIR index
  const int cout0 = mt * BM;
  const int vt0 = ot0 * 2 - 1, vh0 = oh0, vw0 = ow0;                      // window origin
  const int plane = a.plane;

  const float* xbase = a.x + (long)n0 * a.x_nstride;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);
  float* ybase = a.y + (long)n0 * a.y_nstride;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);

  unsigned goff[PCH];
  {
    const int hw = a.WH * a.WW;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned off = OOB;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * a.WW;
        const int n = n0 + wn_;
        const int it = vt0 + wt, ih = vh0 + wh, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && ih < a.Hi && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4);
      }
      goff[j] = off;
    }
  }
  const unsigned wvoff = (unsigned)((((lane * 4) / BM) * a.CoutP + (lane * 4) % BM) * 4);

  // XV4: the window of a temporal stencil has no halo along the flattened (H,W) axis, so with
  // everything 16-byte aligned it can be staged by 16-byte DMA.  The LDS image [c][plane] is
  // then filled in 1 KiB pieces that run across channel boundaries: piece j, lane L holds
  // floats 256j + 4L .. +3.  Wave w owns pieces w, w+4, ...
  constexpr int PV = (CC * PCH * 64 / 256 + 3) / 4;      // pieces per wave (upper bound)
  unsigned xvoff[PV];
  int xvc[PV];
  if (XV4) {
#pragma unroll
    for (int jj = 0; jj < PV; ++jj) {
      const int flat = (wave + 4 * jj) * 256 + lane * 4;
      const int c = flat / plane, e = flat - c * plane;
      unsigned off = OOB;
      if (c < CC) {
        const int wn_ = fdiv(e, a.inv_plane1);
        const int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_ww);              // WH == 1 here
        const int ww = q - wt * a.WW;
        const int n = n0 + wn_, it = vt0 + wt, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + (long)it * a.Wi + iw + (long)c * a.x_cstride) * 4);
      }
      xvoff[jj] = off;
      xvc[jj] = c;
    }
  }

  // pair position of this lane: window offset of its frame d0 (d1..d3 follow at +WH*WW each)
  int lanebase[NF];
  const int fstride = a.WH * a.WW;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNP / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = W_FLOATS + tn * a.plane1 + ((tt * 2) * a.WH + th) * a.WW + tw + half * planeS;
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  f32x16 acc[MF][NF][4];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][t][i] = 0.f;

  auto stage = [&](int cin0, float* sbase) {
    for (int p = wave; p < WPIECES; p += 4) {
      const int row0 = p * RPP;
      const int tap = row0 / CC, c0 = row0 % CC;
      const unsigned soff = (unsigned)((((long)tap * a.CinP + cin0 + c0) * a.CoutP + cout0) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sbase + p * 256), 16, wvoff, soff, 0, 0);
    }
    float* xs = sbase + W_FLOATS;
    if (XV4) {
      const unsigned soff = (unsigned)cin0 * (unsigned)a.x_cstride * 4u;
#pragma unroll
      for (int jj = 0; jj < PV; ++jj) {
        const int j = wave + 4 * jj;
        if (j * 256 < CC * plane && xvc[jj] < CC) {   // exec-masked: lanes past the image write nothing
          const unsigned vo = cin0 + xvc[jj] < a.Cin ? xvoff[jj] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + j * 256), 16, vo, soff, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + c * planeS + j * 64), 4,
                                                     goff[j], soff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) xs[c * planeS + j * 64 + lane] = 0.f;
      }
    }
  };

  const int nchunks = a.nchunks;
  stage(0, smem);
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * stage_floats;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((ch + 1) & 1) * stage_floats);

    constexpr int QS = CC / 2;
    // step q: channel pair (2q, 2q+1); operands of step q+1 are fetched under the MFMAs of q
    auto fetch = [&](int q, float (&av)[MF][4], float (&dv)[NF][4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) av[mf][t] = cur[abase + (t * CC + 2 * q) * BM + mf * 32];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int k = 0; k < 4; ++k) dv[nf][k] = cur[lanebase[nf] + k * fstride + 2 * q * planeS];
    };
    float av[2][MF][4], dv[2][NF][4];
    fetch(0, av[0], dv[0]);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q + 1 < QS) fetch(q + 1, av[(q + 1) & 1], dv[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float d0 = dv[q & 1][nf][0], d1 = dv[q & 1][nf][1], d2 = dv[q & 1][nf][2],
                    d3 = dv[q & 1][nf][3];
        const float D[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            acc[mf][nf][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][mf][t], D[t],
                                                                  acc[mf][nf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: y[2p] = m0+m1+m2, y[2p+1] = m1-m2-m3 ---------------------------------------
  unsigned yvoff[NF];
  bool pvalid[NF], p2valid[NF];
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
  const unsigned frame_bytes = (unsigned)(a.yHf * a.yWf) * 4u;
  const int To_full = a.yst;     // launcher passes the un-paired frame count here
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNP / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, tp = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    pvalid[nf] = n < a.N && 2 * tp < To_full && oh < a.Ho && ow < a.Wo;
    p2valid[nf] = pvalid[nf] && 2 * tp + 1 < To_full;
    const long e = (long)tn * a.y_nstride + ((long)(2 * tp) * a.yHf + oh) * a.yWf + ow;
    yvoff[nf] = pvalid[nf] ? (unsigned)(e * 4) + half_rows : OOB;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;
  if (want_stats) __syncthreads();

  auto emit = [&](auto acc_tag, auto fancy_tag) {       // FANCY: see conv_igemm_body
    constexpr bool ACCUM = decltype(acc_tag)::value;
    constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float s = 0.f, ss = 0.f;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float m0 = acc[mf][nf][0][i], m1 = acc[mf][nf][1][i], m2 = acc[mf][nf][2][i],
                      m3 = acc[mf][nf][3][i];
          float v0 = (m0 + m1) + m2, v1 = (m1 - m2) - m3;
          const unsigned vo0 = cok ? yvoff[nf] : OOB;
          const unsigned vo1 = (cok && p2valid[nf]) ? yvoff[nf] + frame_bytes : OOB;
          if (ACCUM) {
            v0 += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo0, soff, 0));
            v1 += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo1, soff, 0));
          }
          const float u0 = pvalid[nf] ? v0 : 0.f, u1 = p2valid[nf] ? v1 : 0.f;
          s += u0 + u1; ss += u0 * u0 + u1 * u1;
          if (FANCY) {
            v0 = (v0 + bia) * sc + sf;
            v1 = (v1 + bia) * sc + sf;
            if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), ry, vo0, soff, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), ry, vo1, soff, 0);
        }
        if (want_stats) {
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  // see conv_igemm_body: the destination is the dz of a BatchNorm(+ReLU) unit, its backward sums are
  // formed here from the two frames in registers
  auto emit_bwd = [&]() {
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.bwd_y + (long)n0 * a.y_nstride), 0, BUF_RANGE, 0x00020000);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int i0 = 0; i0 < 16; i0 += 4) {
        float yv[4][NF][2];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = i0 + ii;
          const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
          const bool cok = cout0 + rowu + 4 * half < a.Cout;
          const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const unsigned vo0 = cok ? yvoff[nf] : OOB;
            const unsigned vo1 = (cok && p2valid[nf]) ? yvoff[nf] + frame_bytes : OOB;
            yv[ii][nf][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vo0, soff, 0));
            yv[ii][nf][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rb, vo1, soff, 0));
          }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = i0 + ii;
          const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
          const int ml = rowu + 4 * half;
          const int co = cout0 + ml;
          const bool cok = co < a.Cout;
          const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
          float bsc = 0.f, bsf = 0.f, mu = 0.f, is = 0.f;
          if (cok) { bsc = a.bwd_scale[co]; bsf = a.bwd_shift[co]; mu = a.bwd_mean[co]; is = a.bwd_invstd[co]; }
          float s = 0.f, ss = 0.f;
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const float m0 = acc[mf][nf][0][i], m1 = acc[mf][nf][1][i], m2 = acc[mf][nf][2][i],
                        m3 = acc[mf][nf][3][i];
            const float v0 = (m0 + m1) + m2, v1 = (m1 - m2) - m3;
            const unsigned vo0 = cok ? yvoff[nf] : OOB;
            const unsigned vo1 = (cok && p2valid[nf]) ? yvoff[nf] + frame_bytes : OOB;
            const float u0 = yv[ii][nf][0], u1 = yv[ii][nf][1];
            const bool on0 = pvalid[nf] && (!a.bwd_relu || fmaf(u0, bsc, bsf) > 0.f);
            const bool on1 = p2valid[nf] && (!a.bwd_relu || fmaf(u1, bsc, bsf) > 0.f);
            const float g0 = on0 ? v0 : 0.f, g1 = on1 ? v1 : 0.f;
            s += g0 + g1;
            ss += g0 * ((u0 - mu) * is) + g1 * ((u1 - mu) * is);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0), ry, vo0, soff, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1), ry, vo1, soff, 0);
          }
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  const bool fancy = a.bias || a.ep_scale || a.relu;
  if (a.bwd_y) emit_bwd();
  else if (fancy) { if (a.accumulate) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{}); }
  else if (a.accumulate) emit(std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{});

  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[(k * BM + tid) * 2];
          ss += red[(k * BM + tid) * 2 + 1];
        }
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}

template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC = 1>
__global__ void __launch_bounds__(256, OCC)
conv_wino_t_kernel(const ConvArgs a) {
  conv_wino_t_body<CC, BM, BNP, PCH, XV4>(a, (int)blockIdx.x, (int)gridDim.x);
}

// two problems in one launch, see conv_igemm_pair_kernel
template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC = 1>
__global__ void __launch_bounds__(256, OCC)
conv_wino_t_pair_kernel(const ConvArgs a0, const ConvArgs a1, const int nb0) {
  if ((int)blockIdx.x < nb0) conv_wino_t_body<CC, BM, BNP, PCH, XV4>(a0, (int)blockIdx.x, nb0);
  else conv_wino_t_body<CC, BM, BNP, PCH, XV4>(a1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0);
}

struct PairSlot;
template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC>
int wino_t_single(const ConvArgs& a, long blocks, size_t lds, hipStream_t stream) {
  auto kern = conv_wino_t_kernel<CC, BM, BNP, PCH, XV4, OCC>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC>
int wino_t_pair(const ConvArgs& a0, const ConvArgs& a1, long b0, long b1, size_t lds, hipStream_t stream) {
  auto kern = conv_wino_t_pair_kernel<CC, BM, BNP, PCH, XV4, OCC>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)(b0 + b1)), dim3(256), lds, stream, a0, a1, (int)b0);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC = 1>
int launch_wino_t(ConvArgs& a, ConvPlan& p, hipStream_t stream, PairSlot* slot = nullptr);

// ------------------------------------------------------------------------------------
// Temporal (3,1,1) stride-1 convolutions through Winograd F(4,3) along T (coclr_conv_desc.algo = 2).
//
// A QUAD of output frames (4p .. 4p+3) needs input frames d0..d5 = 4p-1 .. 4p+4 and SIX channel
// contractions instead of twelve (F(2,3) above: eight):
//     D0 = 4 d0 - 5 d2 + d4          D1 = (d4 - 4 d2) + (d3 - 4 d1)     D2 = (d4 - 4 d2) - (d3 - 4 d1)
//     D3 = (d4 - d2) + 2 (d3 - d1)   D4 = (d4 - d2) - 2 (d3 - d1)       D5 = 4 d1 - 5 d3 + d5
//     m_i = U_i D_i   (U0 = g0/4, U1 = -(g0+g1+g2)/6, U2 = -(g0-g1+g2)/6, U3 = g0/24 + g1/12 + g2/6,
//                      U4 = g0/24 - g1/12 + g2/6, U5 = g2: [Cout][Cin] matrices, a 6-"tap" packed operand)
//     y[4p]   = m0 + (m1 + m2) + (m3 + m4)        y[4p+1] = (m1 - m2) + 2 (m3 - m4)
//     y[4p+2] = (m1 + m2) + 4 (m3 + m4)           y[4p+3] = (m1 - m2) + 8 (m3 - m4) + m5
// (interpolation points 0, +-1, +-2, inf).  Everything stays fp32; against float64 the result carries ~3.5x the
// rounding error of the direct fp32 convolution (L2 7e-7 against 2e-7 at 192 channels; F(2,3): 2.7e-7) --
// three orders of magnitude inside the 1e-3 the path is held to.
//
// Same structure as conv_wino_t_body: a (6,1,1) stencil with temporal stride 4 over "quad positions", six
// accumulator sets per 32x32 block (96 registers: three workgroups per CU), the epilogue emits four frames.
// 2x fewer MFMAs than the direct form, 1.33x fewer than F(2,3).
template <int CC, int BM, int BNQ, int PCH, bool XV4>
__device__ __forceinline__ void conv_wino_t4_body(const ConvArgs& a, int bid, const int nblocks) {
  constexpr int TAPS = 6;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BNQ / (WN * 32);
  constexpr int RPP = 256 / BM;
  constexpr int WPIECES = TAPS * CC / RPP;
  static_assert(CC % RPP == 0 && CC % 4 == 0, "chunk shape");
  constexpr int W_FLOATS = TAPS * CC * BM;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  if (a.xcd) {                        // XCD-aware tile ids (see conv_igemm_body)
    const int per = nblocks >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const int mt = bid % a.mtiles;
  const int ntile = bid / a.mtiles;
  int r = ntile;
  const int bw_ = r % a.nbw; r /= a.nbw;
  const int bh_ = r % a.nbh; r /= a.nbh;
  const int bt_ = r % a.nbt; r /= a.nbt;
  const int n0 = r << a.lTN;
  const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;   // ot0: QUAD index
  const int cout0 = mt * BM;
  const int vt0 = ot0 * 4 - 1, vh0 = oh0, vw0 = ow0;                      // window origin
  const int plane = a.plane;

  const float* xbase = a.x + (long)n0 * a.x_nstride;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);
  float* ybase = a.y + (long)n0 * a.y_nstride;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);

  unsigned goff[XV4 ? 1 : PCH];
  if (!XV4) {
    const int hw = a.WH * a.WW;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned off = OOB;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * a.WW;
        const int n = n0 + wn_;
        const int it = vt0 + wt, ih = vh0 + wh, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && ih < a.Hi && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4);
      }
      goff[j] = off;
    }
  }
  const unsigned wvoff = (unsigned)((((lane * 4) / BM) * a.CoutP + (lane * 4) % BM) * 4);

  // XV4: 16-byte DMA of the halo-free window, 1 KiB pieces across channel rows (see conv_wino_t_body)
  constexpr int PV = (CC * PCH * 64 / 256 + 3) / 4;
  unsigned xvoff[PV];
  int xvc[PV];
  if (XV4) {
#pragma unroll
    for (int jj = 0; jj < PV; ++jj) {
      const int flat = (wave + 4 * jj) * 256 + lane * 4;
      const int c = flat / plane, e = flat - c * plane;
      unsigned off = OOB;
      if (c < CC) {
        const int wn_ = fdiv(e, a.inv_plane1);
        const int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_ww);              // WH == 1 here
        const int ww = q - wt * a.WW;
        const int n = n0 + wn_, it = vt0 + wt, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + (long)it * a.Wi + iw + (long)c * a.x_cstride) * 4);
      }
      xvoff[jj] = off;
      xvc[jj] = c;
    }
  }

  // quad position of this lane: window offset of its frame d0 (d1..d5 follow at +WH*WW each)
  int lanebase[NF];
  const int fstride = a.WH * a.WW;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = W_FLOATS + tn * a.plane1 + ((tt * 4) * a.WH + th) * a.WW + tw + half * planeS;
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  f32x16 acc[MF][NF][6];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][t][i] = 0.f;

  auto stage = [&](int cin0, float* sbase) {
    for (int p = wave; p < WPIECES; p += 4) {
      const int row0 = p * RPP;
      const int tap = row0 / CC, c0 = row0 % CC;
      const unsigned soff = (unsigned)((((long)tap * a.CinP + cin0 + c0) * a.CoutP + cout0) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sbase + p * 256), 16, wvoff, soff, 0, 0);
    }
    float* xs = sbase + W_FLOATS;
    if (XV4) {
      const unsigned soff = (unsigned)cin0 * (unsigned)a.x_cstride * 4u;
#pragma unroll
      for (int jj = 0; jj < PV; ++jj) {
        const int j = wave + 4 * jj;
        if (j * 256 < CC * plane && xvc[jj] < CC) {   // exec-masked: lanes past the image write nothing
          const unsigned vo = cin0 + xvc[jj] < a.Cin ? xvoff[jj] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + j * 256), 16, vo, soff, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + c * planeS + j * 64), 4,
                                                     goff[XV4 ? 0 : j], soff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) xs[c * planeS + j * 64 + lane] = 0.f;
      }
    }
  };

  const int nchunks = a.nchunks;
  stage(0, smem);
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * stage_floats;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((ch + 1) & 1) * stage_floats);

    constexpr int QS = CC / 2;
    // step q: channel pair (2q, 2q+1); operands of step q+1 are fetched under the MFMAs of q
    auto fetch = [&](int q, float (&av)[MF][6], float (&dv)[NF][6]) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) av[mf][t] = cur[abase + (t * CC + 2 * q) * BM + mf * 32];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int k = 0; k < 6; ++k) dv[nf][k] = cur[lanebase[nf] + k * fstride + 2 * q * planeS];
    };
    float av[2][MF][6], dv[2][NF][6];
    fetch(0, av[0], dv[0]);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q + 1 < QS) fetch(q + 1, av[(q + 1) & 1], dv[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float d0 = dv[q & 1][nf][0], d1 = dv[q & 1][nf][1], d2 = dv[q & 1][nf][2],
                    d3 = dv[q & 1][nf][3], d4 = dv[q & 1][nf][4], d5 = dv[q & 1][nf][5];
        const float t1 = fmaf(-4.f, d2, d4), t2 = fmaf(-4.f, d1, d3);
        const float u1 = d4 - d2, v = d3 - d1;
        const float D[6] = {fmaf(4.f, d0, fmaf(-5.f, d2, d4)), t1 + t2, t1 - t2, fmaf(2.f, v, u1),
                            fmaf(-2.f, v, u1), fmaf(4.f, d1, fmaf(-5.f, d3, d5))};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            acc[mf][nf][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][mf][t], D[t],
                                                                  acc[mf][nf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: four frames per quad position ---------------------------------------------------
  unsigned yvoff[NF];
  int nvalid[NF];                 // valid frames of the quad (0 = position outside the tensor)
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
  // destination lattice along T (one phase of a strided data gradient): output frame o lands on frame
  // o * yst + yot of the destination tensor; dense: yst = 1, yot = 0
  const unsigned frame_bytes = (unsigned)(a.yHf * a.yWf) * 4u * (unsigned)a.yst;
  const int To_full = a.To_full;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, tp = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    const bool ok = n < a.N && 4 * tp < To_full && oh < a.Ho && ow < a.Wo;
    int nv = To_full - 4 * tp;
    nvalid[nf] = ok ? (nv > 4 ? 4 : nv) : 0;
    const long e = (long)tn * a.y_nstride + ((long)(4 * tp * a.yst + a.yot) * a.yHf + oh) * a.yWf + ow;
    yvoff[nf] = ok ? (unsigned)(e * 4) + half_rows : OOB;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;
  if (want_stats) __syncthreads();

  auto emit = [&](auto acc_tag, auto fancy_tag) {       // FANCY: see conv_igemm_body
    constexpr bool ACCUM = decltype(acc_tag)::value;
    constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float s = 0.f, ss = 0.f;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float m0 = acc[mf][nf][0][i], m1 = acc[mf][nf][1][i], m2 = acc[mf][nf][2][i],
                      m3 = acc[mf][nf][3][i], m4 = acc[mf][nf][4][i], m5 = acc[mf][nf][5][i];
          const float sa = m1 + m2, sb = m1 - m2, sc_ = m3 + m4, sd = m3 - m4;
          float v[4] = {(m0 + sa) + sc_, fmaf(2.f, sd, sb), fmaf(4.f, sc_, sa), fmaf(8.f, sd, sb) + m5};
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            const bool fv = f < nvalid[nf];
            const unsigned vo = (cok && fv) ? yvoff[nf] + (unsigned)f * frame_bytes : OOB;
            if (ACCUM) v[f] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo, soff, 0));
            const float u = fv ? v[f] : 0.f;
            s += u; ss += u * u;
            if (FANCY) {
              v[f] = (v[f] + bia) * sc + sf;
              if (a.relu) v[f] = fmaxf(v[f], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[f]), ry, vo, soff, 0);
          }
        }
        if (want_stats) {
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  const bool fancy = a.bias || a.ep_scale || a.relu;
  if (fancy) { if (a.accumulate) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{}); }
  else if (a.accumulate) emit(std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{});

  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[(k * BM + tid) * 2];
          ss += red[(k * BM + tid) * 2 + 1];
        }
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC = 1>
__global__ void __launch_bounds__(256, OCC)
conv_wino_t4_kernel(const ConvArgs a) {
  conv_wino_t4_body<CC, BM, BNQ, PCH, XV4>(a, (int)blockIdx.x, (int)gridDim.x);
}

// two problems in one launch, see conv_igemm_pair_kernel
template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC = 1>
__global__ void __launch_bounds__(256, OCC)
conv_wino_t4_pair_kernel(const ConvArgs a0, const ConvArgs a1, const int nb0) {
  if ((int)blockIdx.x < nb0) conv_wino_t4_body<CC, BM, BNQ, PCH, XV4>(a0, (int)blockIdx.x, nb0);
  else conv_wino_t4_body<CC, BM, BNQ, PCH, XV4>(a1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0);
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC>
int wino_t4_single(const ConvArgs& a, long blocks, size_t lds, hipStream_t stream) {
  auto kern = conv_wino_t4_kernel<CC, BM, BNQ, PCH, XV4, OCC>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC>
int wino_t4_pair(const ConvArgs& a0, const ConvArgs& a1, long b0, long b1, size_t lds, hipStream_t stream) {
  auto kern = conv_wino_t4_pair_kernel<CC, BM, BNQ, PCH, XV4, OCC>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)(b0 + b1)), dim3(256), lds, stream, a0, a1, (int)b0);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC = 1>
int launch_wino_t4(ConvArgs& a, ConvPlan& p, hipStream_t stream, PairSlot* slot = nullptr);

// ------------------------------------------------------------------------------------
// A 4-tap temporal stencil (4,1,1) / stride 1 / pad (1,0,0) through Winograd F(2,4) (coclr_conv_desc.algo = 2
// on that stencil): the odd phase of the strided stem conv's data gradient (ConvGeom.dgrad_phases).  A PAIR of
// output frames (2p, 2p+1) needs input frames o0..o4 = 2p-1 .. 2p+3 and FIVE channel contractions instead of
// eight (points 0, 1, -1, 2, inf; the same transform as the even-tap half of conv_poly7_body):
//     O = (2(o0-o2)+(o3-o1), -2o1-o2+o3, 2o1-3o2+o3, o3-o1, 2(o1-o3)+(o4-o2))
//     U = (b0/2, -(b0+b1+b2+b3)/2, (-b0+b1-b2+b3)/6, (b0+2b1+4b2+8b3)/6, b3)
//     y[2p] = n0+n1+n2+n3        y[2p+1] = n1-n2+2n3+n4
// Writes through the destination lattice along T like conv_wino_t4_body.
template <int CC, int BM, int BNQ, int PCH, bool XV4>
__device__ __forceinline__ void conv_wino_t24_body(const ConvArgs& a, int bid, const int nblocks) {
  constexpr int TAPS = 5;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BNQ / (WN * 32);
  constexpr int RPP = 256 / BM;
  constexpr int WPIECES = TAPS * CC / RPP;
  static_assert(CC % RPP == 0 && CC % 4 == 0, "chunk shape");
  constexpr int W_FLOATS = TAPS * CC * BM;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  if (a.xcd) {                        // XCD-aware tile ids (see conv_igemm_body)
    const int per = nblocks >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const int mt = bid % a.mtiles;
  const int ntile = bid / a.mtiles;
  int r = ntile;
  const int bw_ = r % a.nbw; r /= a.nbw;
  const int bh_ = r % a.nbh; r /= a.nbh;
  const int bt_ = r % a.nbt; r /= a.nbt;
  const int n0 = r << a.lTN;
  const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;   // ot0: PAIR index
  const int cout0 = mt * BM;
  const int vt0 = ot0 * 2 - 1, vh0 = oh0, vw0 = ow0;                      // window origin
  const int plane = a.plane;

  const float* xbase = a.x + (long)n0 * a.x_nstride;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);
  float* ybase = a.y + (long)n0 * a.y_nstride;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);

  unsigned goff[XV4 ? 1 : PCH];
  if (!XV4) {
    const int hw = a.WH * a.WW;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned off = OOB;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * a.WW;
        const int n = n0 + wn_;
        const int it = vt0 + wt, ih = vh0 + wh, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && ih < a.Hi && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4);
      }
      goff[j] = off;
    }
  }
  const unsigned wvoff = (unsigned)((((lane * 4) / BM) * a.CoutP + (lane * 4) % BM) * 4);

  // XV4: 16-byte DMA of the halo-free window, 1 KiB pieces across channel rows (see conv_wino_t_body)
  constexpr int PV = (CC * PCH * 64 / 256 + 3) / 4;
  unsigned xvoff[PV];
  int xvc[PV];
  if (XV4) {
#pragma unroll
    for (int jj = 0; jj < PV; ++jj) {
      const int flat = (wave + 4 * jj) * 256 + lane * 4;
      const int c = flat / plane, e = flat - c * plane;
      unsigned off = OOB;
      if (c < CC) {
        const int wn_ = fdiv(e, a.inv_plane1);
        const int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_ww);              // WH == 1 here
        const int ww = q - wt * a.WW;
        const int n = n0 + wn_, it = vt0 + wt, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + (long)it * a.Wi + iw + (long)c * a.x_cstride) * 4);
      }
      xvoff[jj] = off;
      xvc[jj] = c;
    }
  }

  // pair position of this lane: window offset of its frame o0 (o1..o4 follow at +WH*WW each)
  int lanebase[NF];
  const int fstride = a.WH * a.WW;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = W_FLOATS + tn * a.plane1 + ((tt * 2) * a.WH + th) * a.WW + tw + half * planeS;
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  f32x16 acc[MF][NF][5];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][t][i] = 0.f;

  auto stage = [&](int cin0, float* sbase) {
    for (int p = wave; p < WPIECES; p += 4) {
      const int row0 = p * RPP;
      const int tap = row0 / CC, c0 = row0 % CC;
      const unsigned soff = (unsigned)((((long)tap * a.CinP + cin0 + c0) * a.CoutP + cout0) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sbase + p * 256), 16, wvoff, soff, 0, 0);
    }
    float* xs = sbase + W_FLOATS;
    if (XV4) {
      const unsigned soff = (unsigned)cin0 * (unsigned)a.x_cstride * 4u;
#pragma unroll
      for (int jj = 0; jj < PV; ++jj) {
        const int j = wave + 4 * jj;
        if (j * 256 < CC * plane && xvc[jj] < CC) {   // exec-masked: lanes past the image write nothing
          const unsigned vo = cin0 + xvc[jj] < a.Cin ? xvoff[jj] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + j * 256), 16, vo, soff, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + c * planeS + j * 64), 4,
                                                     goff[XV4 ? 0 : j], soff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) xs[c * planeS + j * 64 + lane] = 0.f;
      }
    }
  };

  const int nchunks = a.nchunks;
  stage(0, smem);
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * stage_floats;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((ch + 1) & 1) * stage_floats);

    constexpr int QS = CC / 2;
    // step q: channel pair (2q, 2q+1); operands of step q+1 are fetched under the MFMAs of q
    auto fetch = [&](int q, float (&av)[MF][5], float (&dv)[NF][5]) {
#pragma unroll
      for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) av[mf][t] = cur[abase + (t * CC + 2 * q) * BM + mf * 32];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int k = 0; k < 5; ++k) dv[nf][k] = cur[lanebase[nf] + k * fstride + 2 * q * planeS];
    };
    float av[2][MF][5], dv[2][NF][5];
    fetch(0, av[0], dv[0]);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q + 1 < QS) fetch(q + 1, av[(q + 1) & 1], dv[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float o0 = dv[q & 1][nf][0], o1 = dv[q & 1][nf][1], o2 = dv[q & 1][nf][2],
                    o3 = dv[q & 1][nf][3], o4 = dv[q & 1][nf][4];
        const float oa = o3 - o2, o31 = o3 - o1;
        const float D[5] = {fmaf(2.f, o0 - o2, o31), fmaf(-2.f, o1, oa), fmaf(2.f, o1, fmaf(-2.f, o2, oa)), o31,
                            fmaf(-2.f, o31, o4 - o2)};
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            acc[mf][nf][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][mf][t], D[t],
                                                                  acc[mf][nf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: two frames per pair position -----------------------------------------------------
  unsigned yvoff[NF];
  int nvalid[NF];                 // valid frames of the quad (0 = position outside the tensor)
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
  // destination lattice along T (one phase of a strided data gradient): output frame o lands on frame
  // o * yst + yot of the destination tensor; dense: yst = 1, yot = 0
  const unsigned frame_bytes = (unsigned)(a.yHf * a.yWf) * 4u * (unsigned)a.yst;
  const int To_full = a.To_full;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, tp = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    const bool ok = n < a.N && 2 * tp < To_full && oh < a.Ho && ow < a.Wo;
    int nv = To_full - 2 * tp;
    nvalid[nf] = ok ? (nv > 2 ? 2 : nv) : 0;
    const long e = (long)tn * a.y_nstride + ((long)(2 * tp * a.yst + a.yot) * a.yHf + oh) * a.yWf + ow;
    yvoff[nf] = ok ? (unsigned)(e * 4) + half_rows : OOB;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;
  if (want_stats) __syncthreads();

  auto emit = [&](auto acc_tag, auto fancy_tag) {       // FANCY: see conv_igemm_body
    constexpr bool ACCUM = decltype(acc_tag)::value;
    constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float s = 0.f, ss = 0.f;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float n0_ = acc[mf][nf][0][i], n1 = acc[mf][nf][1][i], n2 = acc[mf][nf][2][i],
                      n3 = acc[mf][nf][3][i], n4 = acc[mf][nf][4][i];
          float v[2] = {(n0_ + n1) + (n2 + n3), (n1 - n2) + fmaf(2.f, n3, n4)};
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const bool fv = f < nvalid[nf];
            const unsigned vo = (cok && fv) ? yvoff[nf] + (unsigned)f * frame_bytes : OOB;
            if (ACCUM) v[f] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo, soff, 0));
            const float u = fv ? v[f] : 0.f;
            s += u; ss += u * u;
            if (FANCY) {
              v[f] = (v[f] + bia) * sc + sf;
              if (a.relu) v[f] = fmaxf(v[f], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[f]), ry, vo, soff, 0);
          }
        }
        if (want_stats) {
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  const bool fancy = a.bias || a.ep_scale || a.relu;
  if (fancy) { if (a.accumulate) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{}); }
  else if (a.accumulate) emit(std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{});

  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[(k * BM + tid) * 2];
          ss += red[(k * BM + tid) * 2 + 1];
        }
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}


template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC = 1>
__global__ void __launch_bounds__(256, OCC)
conv_wino_t24_kernel(const ConvArgs a) {
  conv_wino_t24_body<CC, BM, BNQ, PCH, XV4>(a, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------------------------
// The temporal stem conv (7,1,1) / stride 2 / pad 3 (STConv3d's second half in Conv_1a, backbone/s3dg.py:41,145)
// in POLYPHASE Winograd form (coclr_conv_desc.algo = 1 on that stencil).
//
// y[t] = sum_k w[k] x[2t + k - 3].  With stride 2 the odd taps (w1, w3, w5) only ever meet one parity of input
// frames and the even taps (w0, w2, w4, w6) the other: for a PAIR of outputs (2p, 2p+1) and the window
// r_j = x[4p - 3 + j], j = 0..8,
//     y[2p]   = (w1 r1 + w3 r3 + w5 r5)  +  (w0 r0 + w2 r2 + w4 r4 + w6 r6)
//     y[2p+1] = (w1 r3 + w3 r5 + w5 r7)  +  (w0 r2 + w2 r4 + w4 r6 + w6 r8)
// i.e. a 3-tap stride-1 correlation over e = (r1, r3, r5, r7) plus a 4-tap one over o = (r0, r2, r4, r6, r8):
// F(2,3) (4 products) + F(2,4) (5 products, points 0, 1, -1, 2, inf) = NINE channel contractions per output
// pair instead of fourteen:
//     E = (e0-e2, e1+e2, e2-e1, e1-e3)                          U_e = (a0, (a0+a1+a2)/2, (a0-a1+a2)/2, a2)
//     O = (2(o0-o2)+(o3-o1), -2o1-o2+o3, 2o1-3o2+o3, o3-o1, 2(o1-o3)+(o4-o2))
//     U_o = (b0/2, -(b0+b1+b2+b3)/2, (-b0+b1-b2+b3)/6, (b0+2b1+4b2+8b3)/6, b3)
//     y[2p] = (m0+m1+m2) + (n0+n1+n2+n3)        y[2p+1] = (m1-m2-m3) + (n1-n2+2n3+n4)
// Same structure as the temporal Winograd kernels above: a (9,1,1) stencil with temporal stride 4 over output-pair
// positions, nine accumulator sets per 32x32 block (144 registers: two workgroups per CU).
template <int CC, int BM, int BNQ, int PCH, bool XV4, bool INAFF = false>
__device__ __forceinline__ void conv_poly7_body(const ConvArgs& a, int bid, const int nblocks) {
  constexpr int TAPS = 9;
  constexpr int WM = 2, WN = 2;
  constexpr int MF = BM / (WM * 32), NF = BNQ / (WN * 32);
  constexpr int RPP = 256 / BM;
  constexpr int WPIECES = TAPS * CC / RPP;
  static_assert(CC % RPP == 0 && CC % 4 == 0, "chunk shape");
  constexpr int W_FLOATS = TAPS * CC * BM;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;

  if (a.xcd) {                        // XCD-aware tile ids (see conv_igemm_body)
    const int per = nblocks >> 3;
    if (bid < (per << 3)) bid = (bid & 7) * per + (bid >> 3);
  }
  const int mt = bid % a.mtiles;
  const int ntile = bid / a.mtiles;
  int r = ntile;
  const int bw_ = r % a.nbw; r /= a.nbw;
  const int bh_ = r % a.nbh; r /= a.nbh;
  const int bt_ = r % a.nbt; r /= a.nbt;
  const int n0 = r << a.lTN;
  const int ow0 = bw_ << a.lTW, oh0 = bh_ << a.lTH, ot0 = bt_ << a.lTT;   // ot0: output-PAIR index
  const int cout0 = mt * BM;
  const int vt0 = ot0 * 4 - 3, vh0 = oh0, vw0 = ow0;                      // window origin: frame 4p - 3
  const int plane = a.plane;

  const float* xbase = a.x + (long)n0 * a.x_nstride;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);
  float* ybase = a.y + (long)n0 * a.y_nstride;
  const __amdgpu_buffer_rsrc_t ry =
      __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);

  unsigned goff[XV4 ? 1 : PCH];
  if (!XV4) {
    const int hw = a.WH * a.WW;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned off = OOB;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * a.WW;
        const int n = n0 + wn_;
        const int it = vt0 + wt, ih = vh0 + wh, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && ih < a.Hi && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4);
      }
      goff[j] = off;
    }
  }
  const unsigned wvoff = (unsigned)((((lane * 4) / BM) * a.CoutP + (lane * 4) % BM) * 4);

  // XV4: 16-byte DMA of the halo-free window, 1 KiB pieces across channel rows (see conv_wino_t_body)
  constexpr int PV = (CC * PCH * 64 / 256 + 3) / 4;
  unsigned xvoff[PV];
  int xvc[PV];
  if (XV4) {
#pragma unroll
    for (int jj = 0; jj < PV; ++jj) {
      const int flat = (wave + 4 * jj) * 256 + lane * 4;
      const int c = flat / plane, e = flat - c * plane;
      unsigned off = OOB;
      if (c < CC) {
        const int wn_ = fdiv(e, a.inv_plane1);
        const int q = e - wn_ * a.plane1;
        const int wt = fdiv(q, a.inv_ww);              // WH == 1 here
        const int ww = q - wt * a.WW;
        const int n = n0 + wn_, it = vt0 + wt, iw = vw0 + ww;
        if (n < a.N && it >= 0 && it < a.Ti && iw < a.Wi)
          off = (unsigned)(((long)wn_ * a.x_nstride + (long)it * a.Wi + iw + (long)c * a.x_cstride) * 4);
      }
      xvoff[jj] = off;
      xvc[jj] = c;
    }
  }

  // pair position of this lane: window offset of its frame r0 (r1..r8 follow at +WH*WW each)
  int lanebase[NF];
  const int fstride = a.WH * a.WW;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = W_FLOATS + tn * a.plane1 + ((tt * 4) * a.WH + th) * a.WW + tw + half * planeS;
  }
  const int abase = half * BM + wm * (BM / WM) + l31;

  // INAFF: the window holds the raw output of the producing BatchNorm unit.  Its per-channel (scale, shift) sit
  // in LDS behind the stages; frames outside [0, Ti) -- r0..r2 of the first output pair, r5..r8 of the last --
  // must stay ZERO after the affine (they are the convolution's zero padding): a 0 / 1 factor per lane
  float* aff = smem + (a.nchunks > 1 ? 2 : 1) * stage_floats;
  float fm[INAFF ? NF : 1][7];
  if (INAFF) {
    for (int c = tid; c < a.CinP; c += 256) {
      aff[c] = c < a.Cin ? a.in_scale[c] : 0.f;
      aff[a.CinP + c] = c < a.Cin ? a.in_shift[c] : 0.f;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int p = wn * (BNQ / WN) + nf * 32 + l31;
      const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
      const int f0 = 4 * (ot0 + tt) - 3;                 // input frame of r0
#pragma unroll
      for (int j = 0; j < 3; ++j) fm[nf][j] = f0 + j >= 0 ? 1.f : 0.f;
      // (r3, r4 feed a valid output frame whenever the pair has one; r5, r6 run past the end only in the last
      // pair of an odd output-frame count)
#pragma unroll
      for (int j = 5; j < 9; ++j) fm[nf][j - 2] = f0 + j < a.Ti ? 1.f : 0.f;
    }
  }

  f32x16 acc[MF][NF][9];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mf][nf][t][i] = 0.f;

  auto stage = [&](int cin0, float* sbase) {
    for (int p = wave; p < WPIECES; p += 4) {
      const int row0 = p * RPP;
      const int tap = row0 / CC, c0 = row0 % CC;
      const unsigned soff = (unsigned)((((long)tap * a.CinP + cin0 + c0) * a.CoutP + cout0) * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(sbase + p * 256), 16, wvoff, soff, 0, 0);
    }
    float* xs = sbase + W_FLOATS;
    if (XV4) {
      const unsigned soff = (unsigned)cin0 * (unsigned)a.x_cstride * 4u;
#pragma unroll
      for (int jj = 0; jj < PV; ++jj) {
        const int j = wave + 4 * jj;
        if (j * 256 < CC * plane && xvc[jj] < CC) {   // exec-masked: lanes past the image write nothing
          const unsigned vo = cin0 + xvc[jj] < a.Cin ? xvoff[jj] : OOB;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + j * 256), 16, vo, soff, 0, 0);
        }
      }
    } else
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + c * planeS + j * 64), 4,
                                                     goff[XV4 ? 0 : j], soff, 0, 0);
      } else {
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) xs[c * planeS + j * 64 + lane] = 0.f;
      }
    }
  };

  const int nchunks = a.nchunks;
  stage(0, smem);
  for (int ch = 0; ch < nchunks; ++ch) {
    const float* cur = smem + (ch & 1) * stage_floats;
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((ch + 1) & 1) * stage_floats);

    constexpr int QS = CC / 2;
    // step q: channel pair (2q, 2q+1); operands of step q+1 are fetched under the MFMAs of q
    auto fetch = [&](int q, float (&av)[MF][9], float (&dv)[NF][9]) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) av[mf][t] = cur[abase + (t * CC + 2 * q) * BM + mf * 32];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int k = 0; k < 9; ++k) dv[nf][k] = cur[lanebase[nf] + k * fstride + 2 * q * planeS];
    };
    float av[2][MF][9], dv[2][NF][9];
    fetch(0, av[0], dv[0]);
#pragma unroll
    for (int q = 0; q < QS; ++q) {
      if (q + 1 < QS) fetch(q + 1, av[(q + 1) & 1], dv[(q + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float* r_ = dv[q & 1][nf];
        if (INAFF) {
          const int c = ch * CC + 2 * q + half;
          const float sc = aff[c], sh = aff[a.CinP + c];
#pragma unroll
          for (int j = 0; j < 9; ++j) {
            float v_ = fmaf(r_[j], sc, sh);
            if (a.in_relu) v_ = fmaxf(v_, 0.f);
            r_[j] = v_;
          }
          r_[0] *= fm[nf][0]; r_[1] *= fm[nf][1]; r_[2] *= fm[nf][2];
          r_[5] *= fm[nf][3]; r_[6] *= fm[nf][4]; r_[7] *= fm[nf][5]; r_[8] *= fm[nf][6];
        }
        // odd taps (w1, w3, w5) on frames r1, r3, r5, r7: F(2,3); even taps (w0, w2, w4, w6) on r0, r2, r4, r6, r8: F(2,4)
        const float e0 = r_[1], e1 = r_[3], e2 = r_[5], e3 = r_[7];
        const float o0 = r_[0], o1 = r_[2], o2 = r_[4], o3 = r_[6], o4 = r_[8];
        const float oa = o3 - o2, o31 = o3 - o1;
        const float D[9] = {e0 - e2, e1 + e2, e2 - e1, e1 - e3,
                            fmaf(2.f, o0 - o2, o31), fmaf(-2.f, o1, oa), fmaf(2.f, o1, fmaf(-2.f, o2, oa)), o31,
                            fmaf(-2.f, o31, o4 - o2)};
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
            acc[mf][nf][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q & 1][mf][t], D[t],
                                                                  acc[mf][nf][t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: two frames per pair position -----------------------------------------------------
  unsigned yvoff[NF];
  int nvalid[NF];                 // valid frames of the pair (0 = position outside the tensor)
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
  const unsigned frame_bytes = (unsigned)(a.yHf * a.yWf) * 4u;
  const int To_full = a.yst;     // launcher passes the un-grouped frame count here
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BNQ / WN) + nf * 32 + l31;
    const int tw = p & ((1 << a.lTW) - 1);
    const int th = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int tt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int tn = p >> (a.lTW + a.lTH + a.lTT);
    const int n = n0 + tn, tp = ot0 + tt, oh = oh0 + th, ow = ow0 + tw;
    const bool ok = n < a.N && 2 * tp < To_full && oh < a.Ho && ow < a.Wo;
    int nv = To_full - 2 * tp;
    nvalid[nf] = ok ? (nv > 2 ? 2 : nv) : 0;
    const long e = (long)tn * a.y_nstride + ((long)(2 * tp) * a.yHf + oh) * a.yWf + ow;
    yvoff[nf] = ok ? (unsigned)(e * 4) + half_rows : OOB;
  }

  const bool want_stats = a.stats != nullptr;
  float* red = smem;
  if (want_stats) __syncthreads();

  auto emit = [&](auto acc_tag, auto fancy_tag) {       // FANCY: see conv_igemm_body
    constexpr bool ACCUM = decltype(acc_tag)::value;
    constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * (BM / WM) + mf * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        const int co = cout0 + ml;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float s = 0.f, ss = 0.f;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float m0 = acc[mf][nf][0][i], m1 = acc[mf][nf][1][i], m2 = acc[mf][nf][2][i],
                      m3 = acc[mf][nf][3][i];
          const float n0_ = acc[mf][nf][4][i], n1 = acc[mf][nf][5][i], n2 = acc[mf][nf][6][i],
                      n3 = acc[mf][nf][7][i], n4 = acc[mf][nf][8][i];
          float v[2] = {((m0 + m1) + m2) + ((n0_ + n1) + (n2 + n3)),
                        ((m1 - m2) - m3) + ((n1 - n2) + fmaf(2.f, n3, n4))};
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const bool fv = f < nvalid[nf];
            const unsigned vo = (cok && fv) ? yvoff[nf] + (unsigned)f * frame_bytes : OOB;
            if (ACCUM) v[f] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo, soff, 0));
            const float u = fv ? v[f] : 0.f;
            s += u; ss += u * u;
            if (FANCY) {
              v[f] = (v[f] + bia) * sc + sf;
              if (a.relu) v[f] = fmaxf(v[f], 0.f);
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[f]), ry, vo, soff, 0);
          }
        }
        if (want_stats) {
          s = row16_sum(s);
          ss = row16_sum(ss);
          if ((lane & 15) == 0) {
            const int slot = wn * 2 + (l31 >> 4);
            red[(slot * BM + ml) * 2 + 0] = s;
            red[(slot * BM + ml) * 2 + 1] = ss;
          }
        }
      }
    }
  };
  const bool fancy = a.bias || a.ep_scale || a.relu;
  if (fancy) { if (a.accumulate) emit(std::true_type{}, std::true_type{}); else emit(std::false_type{}, std::true_type{}); }
  else if (a.accumulate) emit(std::true_type{}, std::false_type{});
  else emit(std::false_type{}, std::false_type{});

  if (want_stats) {
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s += red[(k * BM + tid) * 2];
          ss += red[(k * BM + tid) * 2 + 1];
        }
        a.stats[(long)co * a.ntiles + ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + ntile] = ss;
      }
    }
  }
}


template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC = 1, bool INAFF = false>
__global__ void __launch_bounds__(256, OCC)
conv_poly7_kernel(const ConvArgs a) {
  conv_poly7_body<CC, BM, BNQ, PCH, XV4, INAFF>(a, (int)blockIdx.x, (int)gridDim.x);
}


// ------------------------------------------------------------------------------------
// Spatial (1,3,3) stride-1 pad-1 convolutions through Winograd F(2x2,3x3).
//
// A 2x2 block of outputs needs the 4x4 input patch d starting one row/column before it and
// SIXTEEN channel contractions instead of 36:
//     V = B^T d B        (B^T rows: d0-d2, d1+d2, d2-d1, d1-d3, applied to rows then columns)
//     M[xi] = U[xi] V[xi]   summed over input channels, xi = 4i+j, U = G g G^T (packed operand)
//     Y = A^T M A        (A^T rows: m0+m1+m2, m1-m2-m3)
// 2.25x fewer MFMAs for the (1,3,3) layers -- the largest share of the S3D conv FLOPs -- in the
// forward and data-gradient passes.  fp32 throughout.
//
// Seen from the implicit-GEMM kernel this is a (1,4,4) stencil with stride (1,2,2) over "block
// positions" (Ho/2 x Wo/2 per frame): same box / window / LDS-DMA staging.  A wave owns ONE
// 32(cout) x 32(block positions) MFMA tile and keeps all 16 transform-domain accumulators of it
// (256 registers -> one workgroup per CU); the 4x4 patch is read from the LDS window as eight
// ds_read_b64 and transformed in registers under the MFMAs of the previous step; the epilogue
// applies A^T . A and stores row pairs as 8-byte buffer stores.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One element of an accumulator set, read where the program says so: left to itself the
// compiler copies whole 16-register sets out of the AGPRs, 256 VGPRs at the epilogue's peak.
__device__ __forceinline__ float agpr_read(float x) {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
  return r;
}

// scalar fp32 add / subtract the compiler cannot pair into packed instructions
__device__ __forceinline__ float fadd_s(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fsub_s(float a, float b) {
  float r;
  asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

constexpr int kWinoGrid = 256;    // persistent grid: one workgroup per CU

// X16: the window is staged with 16-BYTE LDS-DMA.  Its rows are widened to whole 16-byte granules of
// the input row (global columns [2*ow0 - 4, 2*ow0 + 2*TW + 4): the patch of block b then starts at the
// odd LDS column 3 + 2b and is read as three aligned 8-byte pairs whose outer halves are unused), every
// lane of a DMA piece moves one granule, and a window of ~100 granules takes TWO pieces per channel
// instead of six 4-byte ones.  Measured at B=32 (same box, alternating): Conv_2c.conv1 0.790 -> 0.781 ms
// forward, 0.686 -> 0.671 data gradient; Mixed_3c.b1.conv1 0.330 -> 0.318 / 0.322 -> 0.307;
// Mixed_3b.b1.conv1 0.175 -> 0.170 / 0.217 -> 0.208.  Needs Wi % 4 == 0 and 16-byte aligned sample /
// channel strides; a.WW, a.plane1 are then the PADDED row / sample extents in floats, a.plane the
// number of granules, a.inv_* the reciprocals in granule units.
template <int CC, int PCH, bool X16 = false, int ABL = 0>
__global__ void __launch_bounds__(256)
conv_wino_hw_kernel(const ConvArgs a, const int total_tiles) {
  constexpr int TAPS = 16;
  constexpr int BM = 64;
  constexpr int RPP = 256 / BM;
  constexpr int WPIECES = TAPS * CC / RPP;
  static_assert(CC % RPP == 0 && CC % 4 == 0, "chunk shape");
  constexpr int W_FLOATS = TAPS * CC * BM;
  constexpr int QS = CC / 2;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;
  // after the two stages, never DMA'd over: per-lane statistics partials [2][BM][64] and the window
  // coordinate table [PCH][256]
  float* redS = smem + 2 * stage_floats;
  float* redQ = redS + BM * 64;
  unsigned* wtab = reinterpret_cast<unsigned*>(redQ + BM * 64);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int plane = a.plane;
  const int WW = a.WW;

  // The workgroup is PERSISTENT: the 16 accumulator sets leave room for one workgroup per CU,
  // so nothing else on the CU would hide its prologue, its epilogue or the launch of its
  // successor, and workgroups marching in lockstep would hit HBM with their stores all at once.
  // It walks tiles L, L+grid, ... and treats their channel chunks as ONE stream through the two
  // LDS stages: the first chunk of the next tile is requested before the epilogue of the current
  // one, and the transformed outputs stay in registers until they are stored under the MFMAs of
  // the next tile's first two chunks.  L keeps the workgroups of one XCD (blockIdx % 8) on
  // neighbouring tiles: the cout tiles of a box share their window through that XCD's L2.
  const int nwg = (int)gridDim.x;
  int tile = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  if (tile >= total_tiles) return;

  const unsigned wvoff = (unsigned)lane * 16u;

  // tile-independent half of the window gather: the window coordinates (n, t, h, w) of the
  // elements (X16: granules; w counts granules) this lane fetches, packed one byte each, parked in LDS
  {
    const int rowlen = X16 ? WW >> 2 : WW;
    const int hw = a.WH * rowlen;
    const int p1 = X16 ? a.plane1 >> 2 : a.plane1;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned crd = 0xffffffffu;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * p1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * rowlen;
        crd = (unsigned)ww | ((unsigned)wh << 8) | ((unsigned)wt << 16) | ((unsigned)wn_ << 24);
      }
      wtab[j * 256 + tid] = crd;
    }
  }

  // block position of this lane inside a box: window offset of the top-left element of its
  // 4x4 patch, and its coordinates
  int lanebase, ptw, pth, ptt, ptn;
  {
    const int p = wn * 32 + l31;
    ptw = p & ((1 << a.lTW) - 1);
    pth = (p >> a.lTW) & ((1 << a.lTH) - 1);
    ptt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    ptn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase = W_FLOATS + ptn * a.plane1 + (ptt * a.WH + pth * 2) * WW + ptw * 2 + half * planeS +
               (X16 ? 3 : 0);
  }
  // weights in LDS: [c][m][16 xi]; the four xi quads of row m sit rotated by m>>2, so the 16-byte
  // reads of 16 neighbouring rows fall on 16 different bank groups
  const int abase = (half * BM + wm * 32 + l31) * 16;
  const int arot = (l31 >> 2) & 3;
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);

  // ---- per-tile state ------------------------------------------------------------------
  int ntile, n0, ot0, oh0, ow0, cout0;        // of the tile being multiplied
  unsigned goff[PCH];
  __amdgpu_buffer_rsrc_t rx;
  auto setup = [&](int t) {
    const int mt = t % a.mtiles;
    ntile = t / a.mtiles;
    int r = ntile;
    const int bw_ = r % a.nbw; r /= a.nbw;
    const int bh_ = r % a.nbh; r /= a.nbh;
    const int bt_ = r % a.nbt; r /= a.nbt;
    n0 = r << a.lTN;
    ow0 = bw_ << a.lTW; oh0 = bh_ << a.lTH; ot0 = bt_ << a.lTT;      // ow0/oh0: BLOCK index
    cout0 = mt * BM;
    const int vt0 = ot0, vh0 = oh0 * 2 - 1, vw0 = ow0 * 2 - (X16 ? 4 : 1);   // window origin
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const unsigned c = wtab[j * 256 + tid];        // read back per tile: cheaper than 10 live registers
      const int ww = (int)(c & 255u), wh = (int)((c >> 8) & 255u), wt = (int)((c >> 16) & 255u),
                wn_ = (int)(c >> 24);
      // X16: a granule starts on a multiple of four columns and Wi % 4 == 0: wholly inside or outside
      const int iw = vw0 + (X16 ? 4 * ww : ww), ih = vh0 + wh, it = vt0 + wt, n = n0 + wn_;
      const bool ok = c != 0xffffffffu && n < a.N && it < a.Ti && (unsigned)ih < (unsigned)a.Hi &&
                      (unsigned)iw < (unsigned)a.Wi;
      const long off = (long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw;
      goff[j] = ok ? (unsigned)off * 4u : OOB;
    }
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)n0 * a.x_nstride), 0, BUF_RANGE,
                                           0x00020000);
  };

  // DMA of one chunk into stage `sbase`.  Weights: the packed operand is [cin][cout][16 xi], so a
  // channel row of the 64-cout tile is 4 KiB = four 1 KiB pieces, copied verbatim.
  auto stage = [&](int cin0, float* sbase) {
    if (!(ABL & 2)) {
      // piece p = wave + 4k is quarter `wave` of channel row k: both offsets advance by constants
      unsigned soff = (unsigned)((((long)cin0 * a.CoutP + cout0) * 16 + wave * 256) * 4);
      const unsigned sstep = (unsigned)a.CoutP * 64u;
      float* dst = sbase + wave * 256;
#pragma unroll
      for (int k = 0; k < WPIECES / 4; ++k) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(dst), 16, wvoff, soff, 0, 0);
        soff += sstep;
        dst += 1024;
      }
    }
    float* xs = sbase + W_FLOATS;
    if (ABL & 1) return;
#pragma unroll
    for (int ci = 0; ci < CC / 4; ++ci) {
      const int c = ci * 4 + wave;
      const int cin = cin0 + c;
      if (cin < a.Cin) {
        const unsigned soff = (unsigned)cin * (unsigned)a.x_cstride * 4u;
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) {
            if (X16) dma16_to_lds(rx, xs + c * planeS + j * 256, goff[j], soff);
            else dma4_to_lds(rx, xs + c * planeS + j * 64, goff[j], soff);
          }
      } else {
#pragma unroll
        for (int j = 0; j < PCH; ++j)
          if (j * 64 < plane) {
            if (X16)
              *reinterpret_cast<float4*>(&xs[c * planeS + j * 256 + lane * 4]) =
                  make_float4(0.f, 0.f, 0.f, 0.f);
            else
              xs[c * planeS + j * 64 + lane] = 0.f;
          }
      }
    }
  };

  const int nchunks = a.nchunks;
  const bool want_stats = a.stats != nullptr;
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
  const unsigned row_bytes = (unsigned)a.yWf * 4u;

  // outputs of the previous tile, waiting to be stored
  float yv[16][4];
  unsigned pend_vo0 = OOB, pend_vo1 = OOB;   // lane offsets of the pending block (rows 2oh, 2oh+1)
  int pend_co0 = 0;                          // first output channel of this wave's 32 rows
  __amdgpu_buffer_rsrc_t pend_ry = rw;
  bool pending = false;
  auto flush = [&](auto lo_tag) {
    constexpr int LO = decltype(lo_tag)::value;
#pragma unroll
    for (int i = LO; i < LO + 8; ++i) {
      // accumulator element i is output row (i&3) + 8*(i>>2), +4 in the upper half-wave
      const int rowu = pend_co0 + (i & 3) + 8 * (i >> 2);
      const unsigned soff = (unsigned)rowu * (unsigned)a.y_cstride * 4u;
      const bool cok = rowu + 4 * half < a.Cout;
      u32x2 s0, s1;
      s0.x = __float_as_uint(yv[i][0]); s0.y = __float_as_uint(yv[i][1]);
      s1.x = __float_as_uint(yv[i][2]); s1.y = __float_as_uint(yv[i][3]);
      __builtin_amdgcn_raw_buffer_store_b64(s0, pend_ry, cok ? pend_vo0 : OOB, soff, 0);
      __builtin_amdgcn_raw_buffer_store_b64(s1, pend_ry, cok ? pend_vo1 : OOB, soff, 0);
    }
  };

  setup(tile);
  stage(0, smem);
  int G = 0;      // chunks streamed so far: chunk G lives in stage G & 1

  for (;;) {
    f32x16 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int ch = 0; ch < nchunks; ++ch, ++G) {
      const float* cur = smem + (G & 1) * stage_floats;
      __builtin_amdgcn_s_waitcnt(0x0f70);   // this wave's DMA of the chunk has landed
      __syncthreads();                      // ... everyone's; the other stage is fully consumed
      if (ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((G + 1) & 1) * stage_floats);
      if (pending) {
        if (ch == 0) flush(std::integral_constant<int, 0>{});
        if (ch == 1 || nchunks == 1) { flush(std::integral_constant<int, 8>{}); pending = false; }
      }

      // step q: channel pair (2q, 2q+1).  Software pipeline with every buffer single except V:
      // while the 16 MFMAs of step q issue, the patch of step q+1 (already in dv) is transformed
      // into V[(q+1)&1], the patch of step q+2 is fetched into dv, and each weight quad of av is
      // refilled for step q+1 as soon as its four MFMAs have gone.
      auto fetch_a_quad = [&](int q, int g, float (&av)[16]) {
        if (ABL & 32) { if (q == 0) { av[4 * g] = av[4 * g + 1] = av[4 * g + 2] = av[4 * g + 3] = 1.f + g; } return; }
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            &cur[abase + 2 * q * BM * 16 + ((g + arot) & 3) * 4]);
        av[4 * g + 0] = v.x; av[4 * g + 1] = v.y; av[4 * g + 2] = v.z; av[4 * g + 3] = v.w;
      };
      // patch rows as two column pairs (the ds_read_b64 granules): the row pass of B^T d B is then
      // elementwise on pairs, the column pass plain scalar ops on their halves
      auto fetch_d = [&](int q, f32x2 (&dv)[8]) {
        if (ABL & 16) {
          if (q == 0) for (int rr = 0; rr < 8; ++rr) { dv[rr].x = 1.f + rr; dv[rr].y = 2.f; }
          return;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float* src = &cur[lanebase + rr * WW + 2 * q * planeS];
          if (X16) {
            // odd column: three aligned 8-byte reads around it (one ds_read2_b64 + one ds_read_b64),
            // the outer halves unused -- 4-byte reads of columns two apart would use every other
            // LDS bank (PMC: 40 % of the LDS cycles conflicts)
            const f32x2 lo = *reinterpret_cast<const f32x2*>(src - 1);
            const f32x2 mid = *reinterpret_cast<const f32x2*>(src + 1);
            const f32x2 hi = *reinterpret_cast<const f32x2*>(src + 3);
            dv[2 * rr].x = lo.y; dv[2 * rr].y = mid.x;
            dv[2 * rr + 1].x = mid.y; dv[2 * rr + 1].y = hi.x;
          } else {
            dv[2 * rr] = *reinterpret_cast<const f32x2*>(src);
            dv[2 * rr + 1] = *reinterpret_cast<const f32x2*>(src + 2);
          }
        }
      };
      auto transform = [&](const f32x2 (&dv)[8], float (&V)[16]) {
        if (ABL & 16) { for (int i = 0; i < 16; ++i) V[i] = dv[i & 7].x; return; }
        f32x2 tl[4], th[4];
        tl[0] = dv[0] - dv[4]; th[0] = dv[1] - dv[5];
        tl[1] = dv[2] + dv[4]; th[1] = dv[3] + dv[5];
        tl[2] = dv[4] - dv[2]; th[2] = dv[5] - dv[3];
        tl[3] = dv[2] - dv[6]; th[3] = dv[3] - dv[7];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          V[4 * i + 0] = tl[i].x - th[i].x;
          V[4 * i + 1] = tl[i].y + th[i].x;
          V[4 * i + 2] = th[i].x - tl[i].y;
          V[4 * i + 3] = tl[i].y - th[i].y;
        }
      };
      float av[16], V[2][16];
      f32x2 dv[8];
      fetch_d(0, dv);
#pragma unroll
      for (int g = 0; g < 4; ++g) fetch_a_quad(0, g, av);
      transform(dv, V[0]);
      if (QS > 1) fetch_d(1, dv);
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < QS) transform(dv, V[(q + 1) & 1]);
        if (q + 2 < QS) fetch_d(q + 2, dv);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            acc[4 * g + k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[4 * g + k], V[q & 1][4 * g + k],
                                                                 acc[4 * g + k], 0, 0, 0);
          if (q + 1 < QS) fetch_a_quad(q + 1, g, av);
        }
        // issue order: one MFMA, two VALU ops of the transform, an LDS read when one is ready
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: Y = A^T M A into registers; hand the LDS stream to the next tile ---------
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long)n0 * a.y_nstride), 0, BUF_RANGE, 0x00020000);
    unsigned yvoff;
    bool pvalid;
    {
      const int n = n0 + ptn, ot = ot0 + ptt, oh = oh0 + pth, ow = ow0 + ptw;   // block coords
      pvalid = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
      const long e = (long)ptn * a.y_nstride + ((long)ot * a.yHf + 2 * oh) * a.yWf + 2 * ow;
      yvoff = pvalid ? (unsigned)(e * 4) + half_rows : OOB;
    }
    const int e_ntile = ntile, e_cout0 = cout0;
    const int next = tile + nwg;
    const bool more = next < total_tiles;
    if (more) {
      setup(next);
      stage(0, smem + (G & 1) * stage_floats);
    }

    // Specialised on what the launch asks for, so the common forms carry no per-element
    // branches: MODE 0 plain, 1 batch-norm statistics (training forward), 2 everything
    // (bias / affine / ReLU / accumulate / statistics decided at run time).
    auto emit = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * 32 + (i & 3) + 8 * (i >> 2);
        const int ml = rowu + 4 * half;
        float r0[4], r1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = agpr_read(acc[0 + j][i]), m1 = agpr_read(acc[4 + j][i]),
                      m2 = agpr_read(acc[8 + j][i]), m3 = agpr_read(acc[12 + j][i]);
          r0[j] = (m0 + m1) + m2;
          r1[j] = (m1 - m2) - m3;
        }
        float v00 = (r0[0] + r0[1]) + r0[2], v01 = (r0[1] - r0[2]) - r0[3];
        float v10 = (r1[0] + r1[1]) + r1[2], v11 = (r1[1] - r1[2]) - r1[3];
        if (MODE == 2) {
          const int co = e_cout0 + ml;
          const bool cok = co < a.Cout;
          if (a.accumulate) {
            const unsigned soff = (unsigned)(e_cout0 + rowu) * (unsigned)a.y_cstride * 4u;
            const unsigned vo0 = cok ? yvoff : OOB;
            const unsigned vo1 = cok && pvalid ? yvoff + row_bytes : OOB;
            const u32x2 o0 = __builtin_amdgcn_raw_buffer_load_b64(ry, vo0, soff, 0);
            const u32x2 o1 = __builtin_amdgcn_raw_buffer_load_b64(ry, vo1, soff, 0);
            v00 += __uint_as_float(o0.x); v01 += __uint_as_float(o0.y);
            v10 += __uint_as_float(o1.x); v11 += __uint_as_float(o1.y);
          }
        }
        if (MODE != 0 && (MODE == 1 || want_stats)) {
          // rows past Cout hold exact zeros (zero weights); blocks past the tensor are masked
          float s = 0.f, ss = 0.f;
          if (pvalid) {
            s = (v00 + v01) + (v10 + v11);
            ss = (v00 * v00 + v01 * v01) + (v10 * v10 + v11 * v11);
          }
          redS[ml * 64 + wn * 32 + l31] = s;      // one partial per block position, summed below
          redQ[ml * 64 + wn * 32 + l31] = ss;
        }
        if (MODE == 2) {
          const int co = e_cout0 + ml;
          const bool cok = co < a.Cout;
          float bia = 0.f, sc = 1.f, sf = 0.f;
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
          v00 = (v00 + bia) * sc + sf; v01 = (v01 + bia) * sc + sf;
          v10 = (v10 + bia) * sc + sf; v11 = (v11 + bia) * sc + sf;
          if (a.relu) {
            v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f);
          }
        }
        yv[i][0] = v00; yv[i][1] = v01; yv[i][2] = v10; yv[i][3] = v11;
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const bool fancy = a.bias || a.ep_scale || a.relu || a.accumulate;
    if (fancy) emit(std::integral_constant<int, 2>{});
    else if (want_stats && !(ABL & 8)) emit(std::integral_constant<int, 1>{});
    else emit(std::integral_constant<int, 0>{});
    pend_ry = ry;
    pend_vo0 = yvoff;
    pend_vo1 = pvalid ? yvoff + row_bytes : OOB;
    pend_co0 = e_cout0 + wm * 32;
    pending = true;

    if (want_stats && !(ABL & 8)) {
      __syncthreads();
      // thread t: row t>>2, quarter t&3 of its 64 partials; the quarters meet through DPP
      const int row = tid >> 2, qtr = tid & 3;
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 ps = *reinterpret_cast<const float4*>(&redS[row * 64 + qtr * 16 + 4 * k]);
        const float4 pq = *reinterpret_cast<const float4*>(&redQ[row * 64 + qtr * 16 + 4 * k]);
        s += (ps.x + ps.y) + (ps.z + ps.w);
        ss += (pq.x + pq.y) + (pq.z + pq.w);
      }
      s += __builtin_amdgcn_update_dpp(0.f, s, 0xB1, 0xf, 0xf, true);
      s += __builtin_amdgcn_update_dpp(0.f, s, 0x4E, 0xf, 0xf, true);
      ss += __builtin_amdgcn_update_dpp(0.f, ss, 0xB1, 0xf, 0xf, true);
      ss += __builtin_amdgcn_update_dpp(0.f, ss, 0x4E, 0xf, 0xf, true);
      const int co = e_cout0 + row;
      if (qtr == 0 && co < a.Cout) {
        a.stats[(long)co * a.ntiles + e_ntile] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + e_ntile] = ss;
      }
    }
    if (!more) break;
    tile = next;
  }
  flush(std::integral_constant<int, 0>{});
  flush(std::integral_constant<int, 8>{});
}

template <int CC, int PCH, bool X16 = false, int ABL = 0>
int launch_wino_hw(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  if (p.WW > 255 || p.WH > 255 || p.WT > 255 || p.lTN > 7) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, 64);
  if (X16) {
    // rows widened to whole 16-byte granules: [2*ow0 - 4, 2*ow0 + 2*TW + 4)
    const int wwp = 2 * (1 << p.lTW) + 8;
    a.WW = wwp;
    a.plane1 = p.WT * p.WH * wwp;
    a.plane = (a.plane1 << p.lTN) / 4;                 // granules
    if (a.plane > PCH * 64) return COCLR_EINVAL;
    a.planeS = cdiv(a.plane, 64) * 256;
    a.inv_plane1 = 1.0f / (float)(a.plane1 / 4);
    a.inv_hw = 1.0f / (float)(p.WH * (wwp / 4));
    a.inv_ww = 1.0f / (float)(wwp / 4);
  } else {
    if (p.plane > PCH * 64) return COCLR_EINVAL;
    a.planeS = cdiv(p.plane, 64) * 64;
  }
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)16 * CC * 64 + (size_t)CC * a.planeS) * sizeof(float);
  // two stages + statistics partials + window coordinate table
  const size_t lds = 2 * stage + (size_t)2 * 64 * 64 * sizeof(float) + (size_t)PCH * 256 * sizeof(unsigned);
  if (lds > 160 * 1024) return COCLR_EINVAL;
  auto kern = conv_wino_hw_kernel<CC, PCH, X16, ABL>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  const long total = (long)a.mtiles * a.ntiles;
  const int grid = total < kWinoGrid ? (int)total : kWinoGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, stream, a, (int)total);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------
// The same F(2x2,3x3) convolution with TWO waves per SIMD (round 4).
//
// What bounds conv_wino_hw_kernel is its single wave per SIMD: 256 accumulators leave room for nothing
// else, and the wave's transform adds, LDS reads and DMA issue run in program order BESIDE its own MFMAs
// (MFMA pipe busy 47 %, section 4.1 of DESIGN.md).  Here the workgroup keeps its 64(cout) x 64(block
// positions) tile, its LDS stages and its DMA traffic, but is EIGHT waves: a wave owns 32 couts x 16 block
// positions as 2 x 16 MFMA tiles of v_mfma_f32_16x16x4_f32 -- 128 accumulators, <= 128 other registers --
// so a SIMD holds two waves and one's transform / LDS / DMA instructions issue under the other's MFMAs.
// Per wave and 4-channel step: one patch transform (lane = block position l&15, channel l>>4), 2 x 16
// weights, 32 MFMAs of 32 cycles -- the same instruction mix per MFMA cycle as the one-wave kernel, the
// window planes 32 floats out of phase so the four channel planes of a read hit different banks.
// 16-byte window DMA only (Wi % 4 == 0); the one-wave kernel keeps every other case.
template <int CC, int PCH, int ABL = 0>
__global__ void __launch_bounds__(512)
conv_wino_hw8_kernel(const ConvArgs a, const int total_tiles) {
  constexpr int TAPS = 16;
  constexpr int BM = 64;
  static_assert(CC == 8, "one window channel per wave");
  constexpr int W_FLOATS = TAPS * CC * BM;
  constexpr int QS = CC / 4;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int stage_floats = W_FLOATS + CC * planeS;
  float* redS = smem + 2 * stage_floats;            // [BM][64] statistics partials, never DMA'd over
  float* redQ = redS + BM * 64;
  unsigned* wtab = reinterpret_cast<unsigned*>(redQ + BM * 64);   // [PCH][64] window coordinates

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, l15 = lane & 15;
  const int wm = wave >> 2, wn = wave & 3;
  const int plane = a.plane;
  const int WW = a.WW;

  const int nwg = (int)gridDim.x;
  int tile = (nwg % 8 == 0) ? ((int)blockIdx.x % 8) * (nwg / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  if (tile >= total_tiles) return;

  const unsigned wvoff = (unsigned)lane * 16u;

  // window coordinates (n, t, h, w-granule) of the granules a lane fetches: the same for every channel
  if (tid < 64) {
    const int rowlen = WW >> 2;
    const int hw = a.WH * rowlen;
    const int p1 = a.plane1 >> 2;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const int e = j * 64 + lane;
      unsigned crd = 0xffffffffu;
      if (e < plane) {
        const int wn_ = fdiv(e, a.inv_plane1);
        int q = e - wn_ * p1;
        const int wt = fdiv(q, a.inv_hw); q -= wt * hw;
        const int wh = fdiv(q, a.inv_ww);
        const int ww = q - wh * rowlen;
        crd = (unsigned)ww | ((unsigned)wh << 8) | ((unsigned)wt << 16) | ((unsigned)wn_ << 24);
      }
      wtab[j * 64 + lane] = crd;
    }
  }
  __syncthreads();

  // block position of this lane inside the box: window offset of the top-left element of its 4x4 patch
  int lanebase;
  {
    const int p = wn * 16 + l15;
    const int ptw = p & ((1 << a.lTW) - 1);
    const int pth = (p >> a.lTW) & ((1 << a.lTH) - 1);
    const int ptt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    const int ptn = p >> (a.lTW + a.lTH + a.lTT);
    lanebase = W_FLOATS + ptn * a.plane1 + (ptt * a.WH + pth * 2) * WW + ptw * 2 + kq * planeS + 3;
  }
  // weights in LDS: [c][m][16 xi], the four xi quads of row m rotated by m>>2 (see the one-wave kernel)
  const int abase = (kq * BM + wm * 32 + l15) * 16;
  const int arot = (l15 >> 2) & 3;
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);

  int ntile, n0, ot0, oh0, ow0, cout0;
  unsigned goff[PCH];
  __amdgpu_buffer_rsrc_t rx;
  auto setup = [&](int t) {
    const int mt = t % a.mtiles;
    ntile = t / a.mtiles;
    int r = ntile;
    const int bw_ = r % a.nbw; r /= a.nbw;
    const int bh_ = r % a.nbh; r /= a.nbh;
    const int bt_ = r % a.nbt; r /= a.nbt;
    n0 = r << a.lTN;
    ow0 = bw_ << a.lTW; oh0 = bh_ << a.lTH; ot0 = bt_ << a.lTT;
    cout0 = mt * BM;
    const int vt0 = ot0, vh0 = oh0 * 2 - 1, vw0 = ow0 * 2 - 4;
#pragma unroll
    for (int j = 0; j < PCH; ++j) {
      const unsigned c = wtab[j * 64 + lane];
      const int ww = (int)(c & 255u), wh = (int)((c >> 8) & 255u), wt = (int)((c >> 16) & 255u),
                wn_ = (int)(c >> 24);
      const int iw = vw0 + 4 * ww, ih = vh0 + wh, it = vt0 + wt, n = n0 + wn_;
      const bool ok = c != 0xffffffffu && n < a.N && it < a.Ti && (unsigned)ih < (unsigned)a.Hi &&
                      (unsigned)iw < (unsigned)a.Wi;
      const long off = (long)wn_ * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw;
      goff[j] = ok ? (unsigned)off * 4u : OOB;
    }
    rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (long)n0 * a.x_nstride), 0, BUF_RANGE,
                                           0x00020000);
  };

  // DMA of one chunk: 32 weight pieces of 1 KiB (channel row k, quarter) -- wave w moves quarter w&3 of
  // rows (w>>2) + 2i -- and window channel `wave`
  auto stage = [&](int cin0, float* sbase) {
    if (ABL & 1) return;      // timing ablation (wrong results): no DMA
    {
      const int k0 = wave >> 2, qtr = wave & 3;
      unsigned soff = (unsigned)(((((long)cin0 + k0) * a.CoutP + cout0) * 16 + qtr * 256) * 4);
      const unsigned sstep = (unsigned)a.CoutP * 128u;
      float* dst = sbase + k0 * 1024 + qtr * 256;
#pragma unroll
      for (int i = 0; i < CC / 2; ++i) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(dst), 16, wvoff, soff, 0, 0);
        soff += sstep;
        dst += 2048;
      }
    }
    // Cin is a multiple of the chunk (the launcher sends everything else to the one-wave kernel): no
    // zero-fill path, whose LDS stores would make the compiler drain the DMA queue in front of them
    float* xs = sbase + W_FLOATS + wave * planeS;
    const unsigned soff = (unsigned)(cin0 + wave) * (unsigned)a.x_cstride * 4u;
#pragma unroll
    for (int j = 0; j < PCH; ++j)
      if (j * 64 < plane) dma16_to_lds(rx, xs + j * 256, goff[j], soff);
  };

  const int nchunks = a.nchunks;
  const bool want_stats = a.stats != nullptr;
  const unsigned row_bytes = (unsigned)a.yWf * 4u;

  // fold the [BM][64] statistics partials of a finished tile: thread t (0..255 of the four waves that do
  // it): row t>>2, quarter t&3 of its 64 partials; the quarters meet through DPP
  auto fold_stats = [&](int t, int f_cout0, int f_ntile) {
    const int row = t >> 2, qtr = t & 3;
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 ps = *reinterpret_cast<const f32x4*>(&redS[row * 64 + qtr * 16 + 4 * k]);
      const f32x4 pq = *reinterpret_cast<const f32x4*>(&redQ[row * 64 + qtr * 16 + 4 * k]);
      s += (ps.x + ps.y) + (ps.z + ps.w);
      ss += (pq.x + pq.y) + (pq.z + pq.w);
    }
    s += __builtin_amdgcn_update_dpp(0.f, s, 0xB1, 0xf, 0xf, true);
    s += __builtin_amdgcn_update_dpp(0.f, s, 0x4E, 0xf, 0xf, true);
    ss += __builtin_amdgcn_update_dpp(0.f, ss, 0xB1, 0xf, 0xf, true);
    ss += __builtin_amdgcn_update_dpp(0.f, ss, 0x4E, 0xf, 0xf, true);
    const int co = f_cout0 + row;
    if (qtr == 0 && co < a.Cout) {
      a.stats[(long)co * a.ntiles + f_ntile] = s;
      a.stats[((long)a.Cout + co) * a.ntiles + f_ntile] = ss;
    }
  };
  // With >= 2 chunks per tile the fold needs no barrier of its own: the partials a tile's epilogue leaves in
  // LDS are folded by waves 4..7 after the NEXT tile's first chunk barrier (every wave has written by then,
  // and the next epilogue writes them again only behind a later barrier).
  const bool fold_later = a.nchunks >= 2;

  // one instantiation of the tile loop per phase (see below): the branch is wave-uniform and both sides
  // pass the same barriers in the same order
  auto run_tiles = [&](auto late_tag) {
  constexpr bool LATE = decltype(late_tag)::value;
  bool stored = false;      // an epilogue has issued its output stores
  bool fold_due = false;    // ... and left statistics partials to fold
  int f_cout0 = 0, f_ntile = 0;
  setup(tile);
  stage(0, smem);
  int G = 0;

  // The two waves of a SIMD (w and w + 4) run A QUARTER CHUNK out of phase: after the per-chunk barrier both
  // would read and transform first and multiply afterwards -- in step, so nothing of one hides under the
  // other.  Waves 4..7 therefore keep the last 16 MFMAs of a chunk back (their operands stay in registers)
  // and issue them after the NEXT barrier, while waves 0..3 read and transform; from then on one wave of
  // the pair multiplies while the other prepares.  Nothing is read from LDS late: the stage being refilled
  // after a barrier was last read before it.
  // Operands: one 16-register weight set whose quads are refilled, as soon as their four MFMAs have issued,
  // with the weights of the next 16-cout half / channel step (12 MFMAs = 384 cycles before their next use).
  for (;;) {
    f32x4 acc[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mb][t][i] = 0.f;
    float av[16], V[16];
    // the 4 MFMAs of weight quad g on accumulator half mb
    auto mma_quad = [&](int mb, int g) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[mb][4 * g + k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4 * g + k], V[4 * g + k],
                                                                 acc[mb][4 * g + k], 0, 0, 0);
    };

    for (int ch = 0; ch < nchunks; ++ch, ++G) {
      const float* cur = smem + (G & 1) * stage_floats;
      // This wave's DMA of the chunk has landed; LDS reads of the other stage are done.  The counter is in
      // issue order, so at the first chunk of a tile the 16 output stores of the previous tile's epilogue
      // -- issued AFTER this chunk's DMA -- may stay in flight (vmcnt(16)); __syncthreads() would drain them.
      if (ch == 0 && stored) __builtin_amdgcn_s_waitcnt(0x4070);
      else __builtin_amdgcn_s_waitcnt(0x0070);
      __builtin_amdgcn_s_barrier();
      if (LATE && ch == 0 && fold_due) { fold_stats(tid - 256, f_cout0, f_ntile); fold_due = false; }
      if (!LATE && ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((G + 1) & 1) * stage_floats);

      auto fetch_a_quad = [&](int q, int mb, int g) {
        if (ABL & 4) { if (q == 0 && mb == 0) { av[4 * g] = av[4 * g + 1] = av[4 * g + 2] = av[4 * g + 3] = 1.f + g; } return; }
        const f32x4 v = *reinterpret_cast<const f32x4*>(
            &cur[abase + mb * 256 + 4 * q * BM * 16 + ((g + arot) & 3) * 4]);
        av[4 * g + 0] = v.x; av[4 * g + 1] = v.y; av[4 * g + 2] = v.z; av[4 * g + 3] = v.w;
      };
      // the 4x4 patch as sixteen scalars d[4 row + col]: the row starts at an odd LDS column, so it comes as
      // three aligned 8-byte reads whose outer halves are unused
      auto fetch_d = [&](int q, float (&dv)[16]) {
        if (ABL & 4) { if (q == 0) for (int rr = 0; rr < 16; ++rr) dv[rr] = 1.f + rr; return; }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float* src = &cur[lanebase + rr * WW + 4 * q * planeS];
          const f32x2 lo = *reinterpret_cast<const f32x2*>(src - 1);
          const f32x2 mid = *reinterpret_cast<const f32x2*>(src + 1);
          const f32x2 hi = *reinterpret_cast<const f32x2*>(src + 3);
          dv[4 * rr + 0] = lo.y; dv[4 * rr + 1] = mid.x; dv[4 * rr + 2] = mid.y; dv[4 * rr + 3] = hi.x;
        }
      };
      // V = B^T d B in 32 SCALAR adds: beside MFMAs a v_pk_add_f32 costs ~13 cycles more than a scalar add
      // (MI355X_MICROARCH.md) and the compiler would pair these up, hence the asm
      auto transform = [&](const float (&dv)[16], float (&V)[16]) {
        if (ABL & 4) { for (int i = 0; i < 16; ++i) V[i] = dv[i]; return; }
        float t[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          t[0 + c] = fsub_s(dv[0 + c], dv[8 + c]);
          t[4 + c] = fadd_s(dv[4 + c], dv[8 + c]);
          t[8 + c] = fsub_s(dv[8 + c], dv[4 + c]);
          t[12 + c] = fsub_s(dv[4 + c], dv[12 + c]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          V[4 * i + 0] = fsub_s(t[4 * i + 0], t[4 * i + 2]);
          V[4 * i + 1] = fadd_s(t[4 * i + 1], t[4 * i + 2]);
          V[4 * i + 2] = fsub_s(t[4 * i + 2], t[4 * i + 1]);
          V[4 * i + 3] = fsub_s(t[4 * i + 1], t[4 * i + 3]);
        }
      };
      // Program order is pinned (sched_barrier): left alone the scheduler sinks every LDS read to just in front
      // of its use -- `ds_read; s_waitcnt lgkmcnt(0)` thirty times per chunk, each a full LDS round trip.
      float dv[16];
      fetch_d(0, dv);
      if (LATE && ch > 0) {
        // the last 16 MFMAs of the previous chunk; their weight quads make room for this chunk's first ones
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          mma_quad(1, g); fetch_a_quad(0, 0, g);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) fetch_a_quad(0, 0, g);
        __builtin_amdgcn_sched_barrier(0);
      }
      // waves 4..7 issue their DMA behind the held-back MFMAs: it runs under the other waves' DMA issue
      if (LATE && ch + 1 < nchunks) stage((ch + 1) * CC, smem + ((G + 1) & 1) * stage_floats);
#pragma unroll
      for (int q = 0; q < QS; ++q) {
        __builtin_amdgcn_sched_barrier(0);
        transform(dv, V);
        if (q + 1 < QS) fetch_d(q + 1, dv);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          mma_quad(0, g); fetch_a_quad(q, 1, g);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (q + 1 < QS) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            mma_quad(1, g); fetch_a_quad(q + 1, 0, g);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if (!LATE) {
#pragma unroll
          for (int g = 0; g < 4; ++g) mma_quad(1, g);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (LATE) {
#pragma unroll
      for (int g = 0; g < 4; ++g) mma_quad(1, g);
    }

    // ---- epilogue: Y = A^T M A into registers; hand the LDS stream to the next tile ---------
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.y + (long)n0 * a.y_nstride), 0, BUF_RANGE, 0x00020000);
    // The lane's position is RE-DERIVED here from the thread index (behind an opaque asm, so the values are
    // not loop-invariant to the compiler): kept live across the chunk loop they are spilled, and a scratch
    // reload in the epilogue is a VMEM load the compiler waits for with vmcnt(0) -- in front of every pair of
    // output stores, which then go out one round trip at a time.
    unsigned yvoff;
    bool pvalid;
    int kq_e, l15_e;
    {
      int lane_e = (int)threadIdx.x & 63;
      asm volatile("" : "+v"(lane_e));
      kq_e = lane_e >> 4; l15_e = lane_e & 15;
      const int p = wn * 16 + l15_e;
      const int ptw = p & ((1 << a.lTW) - 1);
      const int pth = (p >> a.lTW) & ((1 << a.lTH) - 1);
      const int ptt = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
      const int ptn = p >> (a.lTW + a.lTH + a.lTT);
      const int n = n0 + ptn, ot = ot0 + ptt, oh = oh0 + pth, ow = ow0 + ptw;   // block coords
      pvalid = n < a.N && ot < a.To && oh < a.Ho && ow < a.Wo;
      const long e = (long)ptn * a.y_nstride + ((long)ot * a.yHf + 2 * oh) * a.yWf + 2 * ow;
      yvoff = pvalid ? (unsigned)(e * 4) + (unsigned)kq_e * 4u * (unsigned)a.y_cstride * 4u : OOB;
    }
    const int e_ntile = ntile, e_cout0 = cout0;
    const int next = tile + nwg;
    const bool more = next < total_tiles;
    if (more) {
      setup(next);
      stage(0, smem + (G & 1) * stage_floats);
    }
    __builtin_amdgcn_sched_barrier(0);      // the DMA stays in front of the stores below (vmcnt(16) above)
    stored = true;

    auto emit = [&](auto mode_tag) {
      constexpr int MODE = decltype(mode_tag)::value;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int mb = e >> 2, i = e & 3;
        const int rowu = wm * 32 + mb * 16 + i;        // wave-uniform part of the row
        const int ml = rowu + 4 * kq_e;
        float r0[4], r1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = agpr_read(acc[mb][0 + j][i]), m1 = agpr_read(acc[mb][4 + j][i]),
                      m2 = agpr_read(acc[mb][8 + j][i]), m3 = agpr_read(acc[mb][12 + j][i]);
          r0[j] = (m0 + m1) + m2;
          r1[j] = (m1 - m2) - m3;
        }
        float v00 = (r0[0] + r0[1]) + r0[2], v01 = (r0[1] - r0[2]) - r0[3];
        float v10 = (r1[0] + r1[1]) + r1[2], v11 = (r1[1] - r1[2]) - r1[3];
        if (MODE == 2) {
          const int co = e_cout0 + ml;
          const bool cok = co < a.Cout;
          if (a.accumulate) {
            const unsigned soff = (unsigned)(e_cout0 + rowu) * (unsigned)a.y_cstride * 4u;
            const unsigned vo0 = cok ? yvoff : OOB;
            const unsigned vo1 = cok && pvalid ? yvoff + row_bytes : OOB;
            const u32x2 o0 = __builtin_amdgcn_raw_buffer_load_b64(ry, vo0, soff, 0);
            const u32x2 o1 = __builtin_amdgcn_raw_buffer_load_b64(ry, vo1, soff, 0);
            v00 += __uint_as_float(o0.x); v01 += __uint_as_float(o0.y);
            v10 += __uint_as_float(o1.x); v11 += __uint_as_float(o1.y);
          }
        }
        if (MODE != 0 && (MODE == 1 || want_stats)) {
          float s = 0.f, ss = 0.f;
          if (pvalid) {
            s = (v00 + v01) + (v10 + v11);
            ss = (v00 * v00 + v01 * v01) + (v10 * v10 + v11 * v11);
          }
          redS[ml * 64 + wn * 16 + l15_e] = s;
          redQ[ml * 64 + wn * 16 + l15_e] = ss;
        }
        if (MODE == 2) {
          const int co = e_cout0 + ml;
          const bool cok = co < a.Cout;
          float bia = 0.f, sc = 1.f, sf = 0.f;
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
          v00 = (v00 + bia) * sc + sf; v01 = (v01 + bia) * sc + sf;
          v10 = (v10 + bia) * sc + sf; v11 = (v11 + bia) * sc + sf;
          if (a.relu) {
            v00 = fmaxf(v00, 0.f); v01 = fmaxf(v01, 0.f); v10 = fmaxf(v10, 0.f); v11 = fmaxf(v11, 0.f);
          }
        }
        {
          // stored at once: with two waves per SIMD there is no register room to park the outputs
          const unsigned soff = (unsigned)(e_cout0 + rowu) * (unsigned)a.y_cstride * 4u;
          const bool cok = e_cout0 + ml < a.Cout;
          u32x2 s0, s1;
          s0.x = __float_as_uint(v00); s0.y = __float_as_uint(v01);
          s1.x = __float_as_uint(v10); s1.y = __float_as_uint(v11);
          if (!(ABL & 2) || v00 == 1234.5f) {     // timing ablation: no output stores
          __builtin_amdgcn_raw_buffer_store_b64(s0, ry, cok ? yvoff : OOB, soff, 0);
          __builtin_amdgcn_raw_buffer_store_b64(s1, ry, cok && pvalid ? yvoff + row_bytes : OOB, soff, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const bool fancy = a.bias || a.ep_scale || a.relu || a.accumulate;
    if (fancy) emit(std::integral_constant<int, 2>{});
    else if (want_stats) emit(std::integral_constant<int, 1>{});
    else emit(std::integral_constant<int, 0>{});

    if (want_stats) {
      if (fold_later) {
        fold_due = true; f_cout0 = e_cout0; f_ntile = e_ntile;
      } else {
        __builtin_amdgcn_s_waitcnt(0xc07f);   // the partials are in LDS; stores and DMA stay in flight
        __builtin_amdgcn_s_barrier();
        if (tid < 256) fold_stats(tid, e_cout0, e_ntile);
      }
    }
    if (!more) break;
    tile = next;
  }
  if (want_stats && fold_later) {           // the last tile's partials
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    if (LATE && fold_due) fold_stats(tid - 256, f_cout0, f_ntile);
  }
  };
  if (wave >= 4) run_tiles(std::true_type{});
  else run_tiles(std::false_type{});
}

template <int CC, int PCH>
int launch_wino_hw8(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  if (p.WW > 255 || p.WH > 255 || p.WT > 255 || p.lTN > 7 || a.Cin % CC) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, 64);
  const int wwp = 2 * (1 << p.lTW) + 8;
  a.WW = wwp;
  a.plane1 = p.WT * p.WH * wwp;
  a.plane = (a.plane1 << p.lTN) / 4;                 // granules
  if (a.plane > PCH * 64) return COCLR_EINVAL;
  // + 32 floats: the four channel planes a patch read touches sit half the banks apart
  a.planeS = cdiv(a.plane, 64) * 256 + 32;
  a.inv_plane1 = 1.0f / (float)(a.plane1 / 4);
  a.inv_hw = 1.0f / (float)(p.WH * (wwp / 4));
  a.inv_ww = 1.0f / (float)(wwp / 4);
  a.nchunks = a.Cin / CC;
  const size_t stage = ((size_t)16 * CC * 64 + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds = 2 * stage + (size_t)2 * 64 * 64 * sizeof(float) + (size_t)PCH * 64 * sizeof(unsigned);
  if (lds > 160 * 1024) return COCLR_EINVAL;
  auto kern = conv_wino_hw8_kernel<CC, PCH>;
#ifdef COCLR_WINO_ABLATE
  {
    static const int abl = getenv("COCLR_W8_ABL") ? atoi(getenv("COCLR_W8_ABL")) : 0;
    if (abl == 1) kern = conv_wino_hw8_kernel<CC, PCH, 1>;
    if (abl == 2) kern = conv_wino_hw8_kernel<CC, PCH, 2>;
    if (abl == 3) kern = conv_wino_hw8_kernel<CC, PCH, 3>;
    if (abl == 4) kern = conv_wino_hw8_kernel<CC, PCH, 4>;
    if (abl == 7) kern = conv_wino_hw8_kernel<CC, PCH, 7>;
    if (abl) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 160 * 1024);
  }
#endif
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  const long total = (long)a.mtiles * a.ntiles;
  const int grid = total < kWinoGrid ? (int)total : kWinoGrid;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, stream, a, (int)total);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------
// Stem kernel: spatial (1,KH,KW) stencil over a handful of input channels (Cin = 3:
// backbone/s3dg.py:145 Conv_1a.conv1, and the five slices of resnet_2d3d.py:138).
//
// With Cin = 3 the whole reduction is K = 3*49 = 147 products: the chunked kernel above
// would pad it to 4 channels (25 % of the MFMAs multiply zeros) and reload the 50 KB weight
// tile for every 128 output positions.  Here instead
//   * the workgroup is PERSISTENT: the [K][64] weight tile is DMA'd into LDS once and stays,
//     a grid of <= 512 workgroups walks the boxes;
//   * K runs over (channel, kh, kw padded to 8): an MFMA step multiplies two neighbouring kw of
//     one stencil row (12.5 % zero work instead of 25 %), every operand read is base+immediate;
//   * the input window of box b+1 streams in through the LDS-DMA engine while box b is
//     multiplied (two window stages, one barrier per box);
//   * BatchNorm partial sums are carried in registers across boxes: one partial per
//     workgroup instead of one per box.
template <int KH, int KW, int CIN, int PCH>
__global__ void __launch_bounds__(256)
conv_stem_kernel(const ConvArgs a, const int nboxes) {
  // reduction index k = (c, kh, slot): KWP slots per stencil row (kw padded to even; the pad
  // slot has zero weights), so MFMA step (row, q) multiplies slots 2q / 2q+1 and every LDS
  // read of a row is `base + immediate`.
  constexpr int TAPS = KH * KW, KWP = (KW + 1) & ~1, QS = KWP / 2;
  constexpr int NROWS = CIN * KH, KROWS = NROWS * KWP, KSTEPS = NROWS * QS;
  constexpr int BM = 64, BN = 128, NF = 2;
  constexpr int W_FLOATS = ((KROWS * BM + 255) / 256) * 256;
  constexpr int WPIECES = W_FLOATS / 256;

  extern __shared__ __align__(16) float smem[];
  const int planeS = a.planeS;
  const int xstage = CIN * planeS;
  float* Ws = smem;
  float* Xs0 = smem + W_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int cout0 = blockIdx.y * BM;
  const int plane = a.plane;
  const bool gather = a.n_index != nullptr;

  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, BUF_RANGE, 0x00020000);

  // ---- weights, once: LDS row (c, kh, slot) <- packed row (tap*CinP + c), pad slot = 0 ----
  for (int p = wave; p < WPIECES; p += 4) {
    const int row = p * 4 + (lane >> 4);
    const int c = row / (KH * KWP), r2 = row - c * (KH * KWP);
    const int kh = r2 / KWP, slot = r2 - kh * KWP;
    const int tap = kh * KW + slot;
    const unsigned voff =
        (row < KROWS && slot < KW)
            ? (unsigned)((((long)tap * a.CinP + c) * a.CoutP + cout0 + (lane & 15) * 4) * 4)
            : OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, LDS_PTR(Ws + p * 256), 16, voff, 0, 0, 0);
  }

  // ---- box-independent part of the window addresses ------------------------------------
  // The per-box work has to stay tiny (the VALU shares the wave with the MFMA stream):
  // byte offset of window element e for a box at (vt0, vh0) = pre[e] + vt0*frame + vh0*row
  // when the box spans the full output width and samples are not gathered ("fast" form);
  // only the frame / row range checks remain per box.
  const bool fast = a.nbw == 1 && !gather;
  const unsigned rowpitch = (unsigned)a.Wi * 4u, framepitch = (unsigned)(a.Hi * a.Wi) * 4u;
  // wave w moves window pieces w, w+4, w+8, ... (64 elements each) of every channel, so a lane
  // only ever needs the addresses of PW = PCH/4 elements
  constexpr int PW = (PCH + 3) / 4;
  unsigned wcoord[PW];   // wn | wt << 8 | wh << 16 | ww << 24, or ~0 past the window
  unsigned pre[PW];
  {
    const int hw = a.WH * a.WW;
    const unsigned npitch = (unsigned)a.x_nstride * 4u;
#pragma unroll
    for (int jj = 0; jj < PW; ++jj) {
      const int e = (wave + 4 * jj) * 64 + lane;
      unsigned v = 0xffffffffu, pr = OOB;
      if (e < plane) {
        const int q0 = fdiv(e, a.inv_plane1);
        int q = e - q0 * a.plane1;
        const int t = fdiv(q, a.inv_hw); q -= t * hw;
        const int h = fdiv(q, a.inv_ww);
        const int w = q - h * a.WW;
        v = (unsigned)q0 | ((unsigned)t << 8) | ((unsigned)h << 16) | ((unsigned)w << 24);
        const int iw = w - a.pw;
        if ((unsigned)iw < (unsigned)a.Wi)
          pr = (unsigned)iw * 4u + (unsigned)q0 * npitch + (unsigned)t * framepitch +
               (unsigned)h * rowpitch;
      }
      wcoord[jj] = v;
      pre[jj] = pr;
    }
  }
  // frames can leave the tensor only with temporal padding / overhang; samples only when a box
  // spans several of them
  const bool chk_t = a.pt != 0 || (a.nbt << a.lTT) * a.st + (a.WT - 1) - a.pt > a.Ti - 1 + a.st;
  const bool chk_n = a.lTN != 0;

  // per-lane position inside a box (shared by all boxes)
  int lanebase[NF];
  unsigned ylane[NF];
  int ptw[NF], pth[NF], ptt[NF], ptn[NF];
  const unsigned half_rows = (unsigned)half * 4u * (unsigned)a.y_cstride * 4u;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int p = wn * (BN / 2) + nf * 32 + l31;
    ptw[nf] = p & ((1 << a.lTW) - 1);
    pth[nf] = (p >> a.lTW) & ((1 << a.lTH) - 1);
    ptt[nf] = (p >> (a.lTW + a.lTH)) & ((1 << a.lTT) - 1);
    ptn[nf] = p >> (a.lTW + a.lTH + a.lTT);
    lanebase[nf] = ptn[nf] * a.plane1 + ((ptt[nf] * a.st) * a.WH + pth[nf] * a.sh) * a.WW +
                   ptw[nf] * a.sw;
    // destination offset = ylane (position inside the box) + a per-box scalar
    ylane[nf] = (unsigned)(((long)ptn[nf] * a.y_nstride +
                            ((long)(ptt[nf] * a.yst) * a.yHf + pth[nf] * a.ysh) * a.yWf +
                            ptw[nf] * a.ysw) * 4) + half_rows;
  }
  const int abase = half * BM + wm * 32 + l31;

  auto box_origin = [&](int box, int& n0, int& ot0, int& oh0, int& ow0) {
    int r = box;
    const int bw_ = r % a.nbw; r /= a.nbw;
    const int bh_ = r % a.nbh; r /= a.nbh;
    const int bt_ = r % a.nbt; r /= a.nbt;
    n0 = r << a.lTN; ot0 = bt_ << a.lTT; oh0 = bh_ << a.lTH; ow0 = bw_ << a.lTW;
  };

  auto stage_x = [&](int box, float* xs) {
    int n0, ot0, oh0, ow0;
    box_origin(box, n0, ot0, oh0, ow0);
    const int vt0 = ot0 * a.st - a.pt, vh0 = oh0 * a.sh - a.ph, vw0 = ow0 * a.sw - a.pw;
    const float* xbase = gather ? a.x : a.x + (long)n0 * a.x_nstride;
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, BUF_RANGE, 0x00020000);
    unsigned goff[PW];
    if (fast) {
      const unsigned sbox = (unsigned)vt0 * framepitch + (unsigned)vh0 * rowpitch;   // mod 2^32
#pragma unroll
      for (int jj = 0; jj < PW; ++jj) {
        const unsigned wc = wcoord[jj];
        bool ok = (unsigned)(vh0 + (int)((wc >> 16) & 255u)) < (unsigned)a.Hi;
        if (chk_t) ok = ok && (unsigned)(vt0 + (int)((wc >> 8) & 255u)) < (unsigned)a.Ti;
        if (chk_n) ok = ok && n0 + (int)(wc & 255u) < a.N;
        // pre == OOB (padding column / past the window) must stay out of range
        goff[jj] = (ok && pre[jj] != OOB) ? pre[jj] + sbox : OOB;
      }
    } else {
      // gathered samples (the key encoder's shuffled batch): ALL index loads first, unconditionally (clamped),
      // then one wait -- a load inside each element's `if (ok)` is five dependent memory round trips per box
      long ns[PW];
#pragma unroll
      for (int jj = 0; jj < PW; ++jj) {
        const int wn_ = (int)(wcoord[jj] & 255u);
        const int n = n0 + wn_;
        ns[jj] = gather ? (long)a.n_index[n < a.N ? n : a.N - 1] : (long)wn_;
      }
#pragma unroll
      for (int jj = 0; jj < PW; ++jj) {
        const unsigned wc = wcoord[jj];
        const int n = n0 + (int)(wc & 255u);
        const int it = vt0 + (int)((wc >> 8) & 255u);
        const int ih = vh0 + (int)((wc >> 16) & 255u);
        const int iw = vw0 + (int)(wc >> 24);
        const bool ok = wc != 0xffffffffu && n < a.N && (unsigned)it < (unsigned)a.Ti &&
                        (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
        goff[jj] = ok ? (unsigned)((ns[jj] * a.x_nstride + ((long)it * a.Hi + ih) * a.Wi + iw) * 4) : OOB;
      }
    }
#pragma unroll
    for (int jj = 0; jj < PW; ++jj) {
      const int j = wave + 4 * jj;
      if (j * 64 < plane) {
#pragma unroll
        for (int c = 0; c < CIN; ++c)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, LDS_PTR(xs + c * planeS + j * 64), 4,
                                                   goff[jj],
                                                   (unsigned)c * (unsigned)a.x_cstride * 4u, 0, 0);
      }
    }
  };

  float st_s[16], st_q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { st_s[i] = 0.f; st_q[i] = 0.f; }

  int box = blockIdx.x;
  if (box < nboxes) stage_x(box, Xs0);
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): my DMA pieces (weights, first box) landed
  __syncthreads();
  for (int it = 0; box < nboxes; box += gridDim.x, ++it) {
    const float* xs = Xs0 + (it & 1) * xstage;
    if (box + (int)gridDim.x < nboxes) stage_x(box + gridDim.x, Xs0 + ((it + 1) & 1) * xstage);

    f32x16 acc[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nf][i] = 0.f;

    // Rolled, software-pipelined reduction (a fully unrolled 74-step body is ~50 KB of code
    // and thrashes the instruction cache): operands of step s+1 and the window offset of
    // step s+2 are in flight while the MFMAs of step s issue.
    // fully unrolled, operands fetched TWO steps ahead of the MFMAs that consume them
    // (rotating register triple buffer; every read is base + immediate)
    auto fetch = [&](int st, float& av, float (&bv)[NF]) {
      const int row = st / QS, q = st % QS;
      const int c = row / KH, kh = row % KH;
      const int xo = c * planeS + kh * a.WW + 2 * q;          // scalar
      av = Ws[abase + (row * KWP + 2 * q) * BM];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bv[nf] = xs[lanebase[nf] + half + xo];
    };
    float av[3], bv[3][NF];
    fetch(0, av[0], bv[0]);
    fetch(1, av[1], bv[1]);
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) {
      if (st + 2 < KSTEPS) fetch(st + 2, av[(st + 2) % 3], bv[(st + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st % 3], bv[st % 3][nf], acc[nf], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }

    // The window of the next box was requested before the MFMA loop: waiting for it HERE,
    // before this box's stores are issued, keeps the store latency off the critical path
    // (vmcnt counts stores too).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue of this box ------------------------------------------------------
    int n0, ot0, oh0, ow0;
    box_origin(box, n0, ot0, oh0, ow0);
    float* ybase = a.y + (long)n0 * a.y_nstride;
    const __amdgpu_buffer_rsrc_t ry =
        __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, BUF_RANGE, 0x00020000);
    const unsigned ybox = (unsigned)((((long)(ot0 * a.yst + a.yot) * a.yHf + (oh0 * a.ysh + a.yoh)) *
                                          a.yWf + (ow0 * a.ysw + a.yow)) * 4);
    const bool box_full = n0 + (1 << a.lTN) <= a.N && ot0 + (1 << a.lTT) <= a.To &&
                          oh0 + (1 << a.lTH) <= a.Ho && ow0 + (1 << a.lTW) <= a.Wo;
    unsigned yvoff[NF];
    bool pvalid[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      pvalid[nf] = box_full || (n0 + ptn[nf] < a.N && ot0 + ptt[nf] < a.To &&
                                oh0 + pth[nf] < a.Ho && ow0 + ptw[nf] < a.Wo);
      yvoff[nf] = pvalid[nf] ? ylane[nf] + ybox : OOB;
    }
    // The plain form (training forward: statistics, nothing else) must not CONTAIN the conditional coefficient /
    // accumulate loads: with them in the loop the compiler waits vmcnt(0) in front of every store, i.e. for all
    // earlier stores of the box to land -- 32 memory round trips per box.
    auto emit = [&](auto fancy_tag) {
      constexpr bool FANCY = decltype(fancy_tag)::value;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rowu = wm * 32 + (i & 3) + 8 * (i >> 2);
        const int co = cout0 + rowu + 4 * half;
        const bool cok = co < a.Cout;
        const unsigned soff = (unsigned)(cout0 + rowu) * (unsigned)a.y_cstride * 4u;
        float bia = 0.f, sc = 1.f, sf = 0.f;
        if (FANCY) {
          if (a.bias && cok) bia = a.bias[co];
          if (a.ep_scale && cok) { sc = a.ep_scale[co]; sf = a.ep_shift[co]; }
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const unsigned vo = cok ? yvoff[nf] : OOB;
          float v = acc[nf][i];
          if (FANCY && a.accumulate)
            v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, vo, soff, 0));
          const float vm = pvalid[nf] ? v : 0.f;
          st_s[i] += vm; st_q[i] += vm * vm;
          if (FANCY) {
            v += bia;
            v = v * sc + sf;
            if (a.relu) v = fmaxf(v, 0.f);
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, vo, soff, 0);
        }
      }
    };
    if (a.bias || a.ep_scale || a.relu || a.accumulate) emit(std::true_type{});
    else emit(std::false_type{});
    // every wave's share of the next window has landed (waited above) and every wave is done
    // reading this one (its MFMAs have issued): a bare barrier, no memory-counter drain
    asm volatile("s_barrier" ::: "memory");
  }

  // ---- one BatchNorm partial per workgroup ---------------------------------------------
  if (a.stats != nullptr) {
    __syncthreads();
    float* red = Xs0;   // [4][BM][2]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ml = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
      const float s = row16_sum(st_s[i]), q = row16_sum(st_q[i]);
      if ((lane & 15) == 0) {
        const int slot = wn * 2 + (l31 >> 4);
        red[(slot * BM + ml) * 2 + 0] = s;
        red[(slot * BM + ml) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BM) {
      const int co = cout0 + tid;
      if (co < a.Cout) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += red[(k * BM + tid) * 2]; q += red[(k * BM + tid) * 2 + 1]; }
        a.stats[(long)co * a.ntiles + blockIdx.x] = s;
        a.stats[((long)a.Cout + co) * a.ntiles + blockIdx.x] = q;
      }
    }
  }
}

constexpr int kStemGrid = 512;   // persistent workgroups of the stem kernel (2 per CU)

template <int KH, int KW, int CIN, int PCH>
int launch_stem(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  constexpr int KROWS = CIN * KH * ((KW + 1) & ~1);
  constexpr int W_FLOATS = ((KROWS * 64 + 255) / 256) * 256;
  if (p.plane > PCH * 64 || a.Cin != CIN) return COCLR_EINVAL;
  if (p.WT > 255 || p.WH > 255 || p.WW > 255 || (1 << p.lTN) > 255) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, 64);
  a.planeS = cdiv(p.plane, 64) * 64;
  a.nchunks = 1;
  const size_t lds = ((size_t)W_FLOATS + 2 * (size_t)CIN * a.planeS) * sizeof(float);
  if (lds > 160 * 1024) return COCLR_EINVAL;
  auto kern = conv_stem_kernel<KH, KW, CIN, PCH>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)a.ntiles, (unsigned)a.mtiles), dim3(256), lds, stream, a,
                     p.nboxes);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// Weight re-layout:  dst[tap][r][c]  (r < RP rows = reduction channels,
// c < CP = produced channels), zero padded.
//   forward : r = cin,  c = cout, src tap = tap_base + tap*tap_step
//   dgrad   : r = cout, c = cin,  src tap = tap_base + (taps-1-tap)*tap_step (stencil flipped)
struct PackDesc {
  const float* w;
  float* dst;
  int Cout, Cin, taps;
  long co_stride, ci_stride;
  int tap_base, tap_step, RP, CP, transpose, row0, col0;
  // (rows, cols) = extent written per tap: the padded operand when it stands alone, only the
  // real sub-block when it is placed inside a wider (pre-zeroed) operand
  int rows, cols, wino;
};

__device__ __forceinline__ void pack_element(const PackDesc& d, long e) {
  const float* __restrict__ w = d.w;
  const int Cout = d.Cout, Cin = d.Cin, taps = d.taps, transpose = d.transpose;
  const long co_stride = d.co_stride, ci_stride = d.ci_stride;
  const int tap_base = d.tap_base, tap_step = d.tap_step;
  const int c = (int)(e % d.cols);
  const long q = e / d.cols;
  const int rr = (int)(q % d.rows);
  const int tap = (int)(q / d.rows);
  float v = 0.f;
  if (d.wino && taps == 16) {
    // U = G g G^T of a 3x3 spatial stencil (16 transform-domain matrices, xi = 4i+j), G rows
    // (1,0,0), (.5,.5,.5), (.5,-.5,.5), (0,0,1); the data gradient uses the stencil rotated by pi
    const bool ok = transpose ? (rr < Cout && c < Cin) : (rr < Cin && c < Cout);
    if (ok) {
      const float* src = transpose ? w + rr * co_stride + c * ci_stride
                                   : w + c * co_stride + rr * ci_stride;
      float g[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) g[k] = src[tap_base + (transpose ? 8 - k : k) * tap_step];
      const int i = tap >> 2, j = tap & 3;
      float col[3];   // (G g)[i][b]
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
        col[b] = i == 0 ? g0 : i == 1 ? 0.5f * ((g0 + g1) + g2) : i == 2 ? 0.5f * ((g0 - g1) + g2) : g2;
      }
      v = j == 0 ? col[0] : j == 1 ? 0.5f * ((col[0] + col[1]) + col[2])
                          : j == 2 ? 0.5f * ((col[0] - col[1]) + col[2]) : col[2];
    }
  } else if (d.wino && taps == 9) {
    // polyphase Winograd operand of the 7-tap stride-2 temporal stem conv (see conv_poly7_body): matrices 0..3 =
    // F(2,3) of the odd taps (w1, w3, w5), 4..8 = F(2,4) of the even taps (w0, w2, w4, w6)
    if (rr < Cin && c < Cout) {
      const float* src = w + c * co_stride + rr * ci_stride + tap_base;
      const float w0 = src[0], w1 = src[tap_step], w2 = src[2 * tap_step], w3 = src[3 * tap_step],
                  w4 = src[4 * tap_step], w5 = src[5 * tap_step], w6 = src[6 * tap_step];
      const float k6 = 1.0f / 6.0f;
      v = tap == 0 ? w1
        : tap == 1 ? 0.5f * ((w1 + w3) + w5)
        : tap == 2 ? 0.5f * ((w1 - w3) + w5)
        : tap == 3 ? w5
        : tap == 4 ? 0.5f * w0
        : tap == 5 ? -0.5f * ((w0 + w2) + (w4 + w6))
        : tap == 6 ? ((w2 - w0) + (w6 - w4)) * k6
        : tap == 7 ? ((w0 + 2.f * w2) + (4.f * w4 + 8.f * w6)) * k6
        : w6;
    }
  } else if (d.wino && taps == 5) {
    // 5 transformed matrices of a 4-tap temporal stencil, F(2,4) (see conv_wino_t24_body); the data gradient
    // uses the flipped stencil
    const bool ok = transpose ? (rr < Cout && c < Cin) : (rr < Cin && c < Cout);
    if (ok) {
      const float* src = transpose ? w + rr * co_stride + c * ci_stride
                                   : w + c * co_stride + rr * ci_stride;
      float b0 = src[tap_base], b1 = src[tap_base + tap_step], b2 = src[tap_base + 2 * tap_step],
            b3 = src[tap_base + 3 * tap_step];
      if (transpose) { float t_ = b0; b0 = b3; b3 = t_; t_ = b1; b1 = b2; b2 = t_; }
      const float k6 = 1.0f / 6.0f;
      v = tap == 0 ? 0.5f * b0
        : tap == 1 ? -0.5f * ((b0 + b1) + (b2 + b3))
        : tap == 2 ? ((b1 - b0) + (b3 - b2)) * k6
        : tap == 3 ? ((b0 + 2.f * b1) + (4.f * b2 + 8.f * b3)) * k6
        : b3;
    }
  } else if (d.wino && taps == 6) {
    // 6 transformed matrices of a 3-tap temporal stencil, F(4,3) (see conv_wino_t4_body); the data
    // gradient uses the flipped stencil
    const bool ok = transpose ? (rr < Cout && c < Cin) : (rr < Cin && c < Cout);
    if (ok) {
      const float* src = transpose ? w + rr * co_stride + c * ci_stride
                                   : w + c * co_stride + rr * ci_stride;
      float w0 = src[tap_base], w1 = src[tap_base + tap_step], w2 = src[tap_base + 2 * tap_step];
      if (transpose) { const float t_ = w0; w0 = w2; w2 = t_; }
      const float k4 = 0.25f, k6 = 1.0f / 6.0f, k12 = 1.0f / 12.0f, k24 = 1.0f / 24.0f;
      v = tap == 0 ? w0 * k4
        : tap == 1 ? -((w0 + w1) + w2) * k6
        : tap == 2 ? -((w0 - w1) + w2) * k6
        : tap == 3 ? (w0 * k24 + w1 * k12) + w2 * k6
        : tap == 4 ? (w0 * k24 - w1 * k12) + w2 * k6
        : w2;
    }
  } else if (d.wino) {
    // 4 transformed matrices of a 3-tap temporal stencil (taps == 4 here): G0 = w0,
    // G1 = (w0+w1+w2)/2, G2 = (w0-w1+w2)/2, G3 = w2; the data gradient uses the flipped stencil
    const bool ok = transpose ? (rr < Cout && c < Cin) : (rr < Cin && c < Cout);
    if (ok) {
      const float* src = transpose ? w + rr * co_stride + c * ci_stride
                                   : w + c * co_stride + rr * ci_stride;
      float w0 = src[tap_base], w1 = src[tap_base + tap_step], w2 = src[tap_base + 2 * tap_step];
      if (transpose) { const float t_ = w0; w0 = w2; w2 = t_; }
      v = tap == 0 ? w0 : tap == 1 ? 0.5f * ((w0 + w1) + w2) : tap == 2 ? 0.5f * ((w0 - w1) + w2) : w2;
    }
  } else if (!transpose) {
    if (rr < Cin && c < Cout) v = w[c * co_stride + rr * ci_stride + tap_base + tap * tap_step];
  } else {
    if (rr < Cout && c < Cin)
      v = w[rr * co_stride + c * ci_stride + tap_base + (taps - 1 - tap) * tap_step];
  }
  if (d.wino && taps == 16)
    // F(2x2,3x3) operand: [r][c][16 xi], the xi quads of column c rotated by c>>2 (the kernel's
    // 16-byte LDS reads of neighbouring columns then hit different bank groups)
    d.dst[((long)(d.row0 + rr) * d.CP + d.col0 + c) * 16 + ((((tap >> 2) + (c >> 2)) & 3) << 2) + (tap & 3)] = v;
  else
    d.dst[((long)tap * d.RP + d.row0 + rr) * d.CP + d.col0 + c] = v;
}

__global__ void pack_weights_kernel(const PackDesc d) {
  const long total = (long)d.taps * d.rows * d.cols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long)gridDim.x * blockDim.x)
    pack_element(d, e);
}

// Every operand of an encoder in one launch: block b works on 1024 elements of table entry
// blockmap[b][0] starting at element 1024 * blockmap[b][1].  The table rows are the 16 int64 that
// coclr_conv_pack_describe writes.
constexpr int kPackBlockElems = 1024;

__global__ void __launch_bounds__(256)
pack_weights_batch_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ blockmap) {
  const int entry = blockmap[2 * blockIdx.x], local = blockmap[2 * blockIdx.x + 1];
  const int64_t* t = table + (long)entry * 16;
  PackDesc d;
  d.w = reinterpret_cast<const float*>(t[0]);
  d.dst = reinterpret_cast<float*>(t[1]);
  d.Cout = (int)t[2]; d.Cin = (int)t[3]; d.taps = (int)t[4];
  d.co_stride = t[5]; d.ci_stride = t[6];
  d.tap_base = (int)t[7]; d.tap_step = (int)t[8]; d.RP = (int)t[9]; d.CP = (int)t[10];
  d.transpose = (int)t[11]; d.row0 = (int)t[12]; d.col0 = (int)t[13];
  d.rows = (int)(t[14] & 0xffffffff); d.cols = (int)(t[14] >> 32); d.wino = (int)t[15];
  const long total = (long)d.taps * d.rows * d.cols;
  const long e0 = (long)local * kPackBlockElems + threadIdx.x;
#pragma unroll
  for (int k = 0; k < kPackBlockElems / 256; ++k) {
    const long e = e0 + k * 256;
    if (e < total) pack_element(d, e);
  }
}

inline int pad_to(int v, int m) { return ((v + m - 1) / m) * m; }

// granules of the widened window of a kw = 3 / stride-1 stencil (see XG above)
inline int granule_count(const ConvPlan& p) {
  return ((p.WT * p.WH * ((1 << p.lTW) + 8)) << p.lTN) / 4;
}

// A launch that coclr_conv3d_fwd_multi may fuse with its neighbour: the launcher fills the slot instead of
// launching; two slots with the same `pair` function are the same kernel variant.
typedef int (*SingleFn)(const ConvArgs&, long, size_t, hipStream_t);
typedef int (*PairFn)(const ConvArgs&, const ConvArgs&, long, long, size_t, hipStream_t);
struct PairSlot {
  ConvArgs args;
  long blocks;
  size_t lds;
  SingleFn single;
  PairFn pair;
  bool pending;
};

template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4, bool XG>
int variant_single(const ConvArgs& a, long blocks, size_t lds, hipStream_t stream) {
  auto kern = conv_igemm_kernel<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4, bool XG>
int variant_pair(const ConvArgs& a0, const ConvArgs& a1, long b0, long b1, size_t lds, hipStream_t stream) {
  auto kern = conv_igemm_pair_kernel<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)(b0 + b1)), dim3(256), lds, stream, a0, a1, (int)b0);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// Two DIFFERENT variants of the (1,3,3) small-map kernel in one launch: on the first 8x8x8 blocks the wide
// branch takes 64x128 tiles and the narrow one 64x64 (choose_tile); each problem keeps its own plan, so
// outputs and statistics are bit-identical to the two single launches.
template <int BNA, int BNB>
__global__ void __launch_bounds__(256)
conv_igemm_133_mixed_kernel(const ConvArgs a0, const ConvArgs a1, const int nb0) {
  if ((int)blockIdx.x < nb0)
    conv_igemm_body<1, 3, 3, 8, 64, BNA, 3, false, true>(a0, (int)blockIdx.x, nb0);
  else
    conv_igemm_body<1, 3, 3, 8, 64, BNB, 3, false, true>(a1, (int)blockIdx.x - nb0, (int)gridDim.x - nb0);
}

template <int BNA, int BNB>
int mixed_pair_launch(const ConvArgs& a0, const ConvArgs& a1, long b0, long b1, size_t lds, hipStream_t stream) {
  auto kern = conv_igemm_133_mixed_kernel<BNA, BNB>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)(b0 + b1)), dim3(256), lds, stream, a0, a1, (int)b0);
  COCLR_LAUNCH_CHECK();
  return 0;
}

inline PairFn mixed_pair(SingleFn f0, SingleFn f1) {
  const SingleFn wide = &variant_single<1, 3, 3, 8, 64, 128, 3, false, true>;
  const SingleFn narrow = &variant_single<1, 3, 3, 8, 64, 64, 3, false, true>;
  if (f0 == wide && f1 == narrow) return &mixed_pair_launch<128, 64>;
  if (f0 == narrow && f1 == wide) return &mixed_pair_launch<64, 128>;
  return nullptr;
}

template <int KT, int KH, int KW, int CC, int BM, int BN, int PCH, bool XV4 = false, bool XG = false>
int launch_variant(ConvArgs& a, ConvPlan& p, hipStream_t stream, PairSlot* slot = nullptr) {
  constexpr int TAPS = KT * KH * KW;
  a.mtiles = cdiv(a.Cout, BM);
  if (XG) {
    const int wwp = (1 << p.lTW) + 8;
    a.WW = wwp;
    a.plane1 = p.WT * p.WH * wwp;
    a.plane = (a.plane1 << p.lTN) / 4;
    if (a.plane > PCH * 64) return COCLR_EINVAL;
    a.planeS = cdiv(a.plane, 64) * 256;
    a.inv_plane1 = 1.0f / (float)(a.plane1 / 4);
    a.inv_hw = 1.0f / (float)(p.WH * (wwp / 4));
    a.inv_ww = 1.0f / (float)(wwp / 4);
  } else {
    if (p.plane > PCH * 64) return COCLR_EINVAL;
    a.planeS = XV4 ? p.plane : cdiv(p.plane, 64) * 64;    // 16-byte staging packs rows back to back
  }
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)TAPS * CC * BM + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds_main = stage * (a.nchunks > 1 ? 2 : 1);
  const size_t lds_red = (size_t)4 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  const long blocks = (long)a.mtiles * a.ntiles;
  // pair kernels exist for the stencils sibling units of an inception block share: (1,3,3) of the two
  // separable branches, 16-byte-staged (1,1,1) of the fused heads and the pool branch
  constexpr bool PAIRABLE = KT == 1 && ((KH == 3 && KW == 3) || (KH == 1 && KW == 1 && XV4));
  if (slot && PAIRABLE) {
    slot->args = a; slot->blocks = blocks; slot->lds = lds; slot->pending = true;
    slot->single = &variant_single<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>;
    if constexpr (PAIRABLE) slot->pair = &variant_pair<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>;
    return 0;
  }
  return variant_single<KT, KH, KW, CC, BM, BN, PCH, XV4, XG>(a, blocks, lds, stream);
}

template <int CC, int BM, int BNP, int PCH, bool XV4, int OCC>
int launch_wino_t(ConvArgs& a, ConvPlan& p, hipStream_t stream, PairSlot* slot) {
  if (p.plane > PCH * 64) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, BM);
  // 16-byte staging packs the channel rows back to back
  a.planeS = XV4 ? p.plane : cdiv(p.plane, 64) * 64;
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)4 * CC * BM + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds_main = stage * (a.nchunks > 1 ? 2 : 1);
  const size_t lds_red = (size_t)4 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  const long blocks = (long)a.mtiles * a.ntiles;
  if (slot) {
    slot->args = a; slot->blocks = blocks; slot->lds = lds; slot->pending = true;
    slot->single = &wino_t_single<CC, BM, BNP, PCH, XV4, OCC>;
    slot->pair = &wino_t_pair<CC, BM, BNP, PCH, XV4, OCC>;
    return 0;
  }
  return wino_t_single<CC, BM, BNP, PCH, XV4, OCC>(a, blocks, lds, stream);
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC>
int launch_wino_t4(ConvArgs& a, ConvPlan& p, hipStream_t stream, PairSlot* slot) {
  if (p.plane > PCH * 64) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, BM);
  a.planeS = XV4 ? p.plane : cdiv(p.plane, 64) * 64;
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)6 * CC * BM + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds_main = stage * (a.nchunks > 1 ? 2 : 1);
  const size_t lds_red = (size_t)4 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  const long blocks = (long)a.mtiles * a.ntiles;
  if (slot) {
    slot->args = a; slot->blocks = blocks; slot->lds = lds; slot->pending = true;
    slot->single = &wino_t4_single<CC, BM, BNQ, PCH, XV4, OCC>;
    slot->pair = &wino_t4_pair<CC, BM, BNQ, PCH, XV4, OCC>;
    return 0;
  }
  return wino_t4_single<CC, BM, BNQ, PCH, XV4, OCC>(a, blocks, lds, stream);
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC>
int launch_wino_t24(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  if (p.plane > PCH * 64) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, BM);
  a.planeS = XV4 ? p.plane : cdiv(p.plane, 64) * 64;
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)5 * CC * BM + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds_main = stage * (a.nchunks > 1 ? 2 : 1);
  const size_t lds_red = (size_t)4 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  const long blocks = (long)a.mtiles * a.ntiles;
  auto kern = conv_wino_t24_kernel<CC, BM, BNQ, PCH, XV4, OCC>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

template <int CC, int BM, int BNQ, int PCH, bool XV4, int OCC, bool INAFF = false>
int launch_poly7(ConvArgs& a, ConvPlan& p, hipStream_t stream) {
  if (p.plane > PCH * 64) return COCLR_EINVAL;
  a.mtiles = cdiv(a.Cout, BM);
  a.planeS = XV4 ? p.plane : cdiv(p.plane, 64) * 64;
  a.nchunks = cdiv(a.Cin, CC);
  const size_t stage = ((size_t)9 * CC * BM + (size_t)CC * a.planeS) * sizeof(float);
  const size_t lds_main = stage * (a.nchunks > 1 ? 2 : 1) + (INAFF ? (size_t)2 * a.CinP * sizeof(float) : 0);
  const size_t lds_red = (size_t)4 * BM * 2 * sizeof(float);
  const size_t lds = lds_main > lds_red ? lds_main : lds_red;
  if (lds > 160 * 1024) return COCLR_EINVAL;
  const long blocks = (long)a.mtiles * a.ntiles;
  auto kern = conv_poly7_kernel<CC, BM, BNQ, PCH, XV4, OCC, INAFF>;
  static std::atomic<uint64_t> attr_done{0};
  COCLR_RETURN_IF(ensure_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024, attr_done));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, stream, a);
  COCLR_LAUNCH_CHECK();
  return 0;
}

// efficiency of covering Cout with tiles of BM rows
inline double cover(int cout, int bm) { return (double)cout / ((double)cdiv(cout, bm) * bm); }

struct Choice { int bm, lbn; };

// Decide (BM, BN) for a stencil class given which variants exist.
//
// Cost model: the matrix pipes are the bound, so a launch takes (workgroups that land on the
// busiest CU) x (MFMAs per workgroup).  When the whole grid is co-resident (<= 6 workgroups
// per CU) the busiest CU holds ceil(nWG / 256) of them -- 640 tiles of 64x128 cost 3 rounds
// although the average is 2.5 -- so the tile that divides the layer evenly wins even if it is
// smaller; 64x64 tiles pay ~10 % for twice the operand traffic per MFMA, and a grid that
// leaves one workgroup per CU pays ~20 % (measured on the 8x8x8 stage, profiles/r01_layers_*).
Choice choose_tile(const ConvPlan& base, int kt, int kh, int kw, bool has128x128, bool has64x64,
                   int max_plane_128, int max_plane_64) {
  struct Cand { int bm, lbn, max_plane; double eff; bool have; };
  const Cand cands[3] = {{128, 7, max_plane_128, 1.00, has128x128},
                         {64, 7, max_plane_128, 1.00, true},
                         {64, 6, max_plane_64, 0.90, has64x64}};
  Choice best{64, 7};
  double best_cost = -1.0;
  for (const Cand& cd : cands) {
    if (!cd.have) continue;
    ConvPlan p = base;
    conv_pick_box(&p, cd.lbn, kt, kh, kw);
    if (p.plane > cd.max_plane) continue;
    const double nwg = (double)p.ntiles * cdiv(base.Cout, cd.bm);
    const double per_cu = nwg <= 256.0 * 6 ? (double)cdiv((long)nwg, 256) : nwg / 256.0;
    // a lone workgroup per CU (one wave per SIMD) cannot hide its own staging / epilogue
    const double occ_eff = per_cu < 1.5 ? 0.80 : (per_cu < 2.5 ? 0.95 : 1.0);
    const double cost = per_cu * (cd.bm / 64) * ((1 << cd.lbn) / 64) / (cd.eff * occ_eff);
    if (best_cost < 0 || cost < best_cost - 1e-9) {   // ties keep the larger tile (listed first)
      best_cost = cost;
      best.bm = cd.bm; best.lbn = cd.lbn;
    }
  }
  return best;
}

}  // namespace

extern "C" int coclr_conv_packed_size(int cin, int cout, int taps, int transpose, int64_t* elems) {
  transpose &= 1;
  const int r = transpose ? cout : cin, c = transpose ? cin : cout;
  const long RP = pad_to(r, 32), CP = pad_to(c, 128);
  *elems = (int64_t)taps * RP * CP;
  return 0;
}

namespace {

// Validate one pack request and express it as a PackDesc.
int pack_describe(const float* w, float* packed, int cout, int cin, int taps, int64_t co_stride,
                  int64_t ci_stride, int tap_base, int tap_step, int transpose, int row0,
                  int rows_total, int col0, int cols_total, PackDesc* d) {
  // bit 1: Winograd operand -- taps = 4: F(2,3) of a 3-tap temporal stencil, taps = 16:
  // F(2x2,3x3) of a 9-tap spatial stencil
  const int wino = (transpose >> 1) & 1;
  transpose &= 1;
  if (cout <= 0 || cin <= 0 || taps <= 0) return COCLR_EINVAL;
  if (wino && taps != 4 && taps != 5 && taps != 6 && taps != 9 && taps != 16) return COCLR_EINVAL;
  if (wino && taps == 9 && transpose) return COCLR_EINVAL;      // forward operand only (the data gradient runs in phases)
  if (wino && taps == 16 && rows_total > 0 && cols_total > 0) return COCLR_EINVAL;   // stand-alone only
  const int r = transpose ? cout : cin, c = transpose ? cin : cout;
  const bool placed = rows_total > 0 && cols_total > 0;
  if (placed && (row0 < 0 || col0 < 0 || row0 + r > rows_total || col0 + c > cols_total))
    return COCLR_EINVAL;
  d->w = w; d->dst = packed;
  d->Cout = cout; d->Cin = cin; d->taps = taps;
  d->co_stride = (long)co_stride; d->ci_stride = (long)ci_stride;
  d->tap_base = tap_base; d->tap_step = tap_step;
  d->RP = pad_to(placed ? rows_total : r, 32);
  d->CP = pad_to(placed ? cols_total : c, 128);
  d->transpose = transpose;
  d->row0 = placed ? row0 : 0; d->col0 = placed ? col0 : 0;
  d->rows = placed ? r : d->RP; d->cols = placed ? c : d->CP;
  d->wino = wino;
  return 0;
}

}  // namespace

extern "C" int coclr_conv_pack_weights(const float* w, float* packed, int cout, int cin, int taps,
                                       int64_t co_stride, int64_t ci_stride, int tap_base,
                                       int tap_step, int transpose, int row0, int rows_total,
                                       int col0, int cols_total, void* stream) {
  PackDesc d;
  int rc = pack_describe(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, tap_step,
                         transpose, row0, rows_total, col0, cols_total, &d);
  if (rc) return rc;
  const long total = (long)d.taps * d.rows * d.cols;
  int blocks = cdiv(total, 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d);
  COCLR_LAUNCH_CHECK();
  return 0;
}

extern "C" int coclr_conv_pack_describe(const float* w, float* packed, int cout, int cin, int taps,
                                        int64_t co_stride, int64_t ci_stride, int tap_base,
                                        int tap_step, int transpose, int row0, int rows_total,
                                        int col0, int cols_total, int64_t* entry,
                                        int32_t* nblocks) {
  PackDesc d;
  int rc = pack_describe(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, tap_step,
                         transpose, row0, rows_total, col0, cols_total, &d);
  if (rc) return rc;
  if (!entry || !nblocks) return COCLR_EINVAL;
  entry[0] = (int64_t)(uintptr_t)d.w; entry[1] = (int64_t)(uintptr_t)d.dst;
  entry[2] = d.Cout; entry[3] = d.Cin; entry[4] = d.taps;
  entry[5] = d.co_stride; entry[6] = d.ci_stride;
  entry[7] = d.tap_base; entry[8] = d.tap_step; entry[9] = d.RP; entry[10] = d.CP;
  entry[11] = d.transpose; entry[12] = d.row0; entry[13] = d.col0;
  entry[14] = (int64_t)(uint32_t)d.rows | ((int64_t)d.cols << 32);
  entry[15] = d.wino;
  *nblocks = (int32_t)cdiv((long)d.taps * d.rows * d.cols, (long)kPackBlockElems);
  return 0;
}

extern "C" int coclr_conv_pack_batch(const int64_t* table, const int32_t* blockmap, int nblocks,
                                     void* stream) {
  if (nblocks < 0 || (nblocks > 0 && (!table || !blockmap))) return COCLR_EINVAL;
  if (nblocks == 0) return 0;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)nblocks), dim3(256), 0,
                     (hipStream_t)stream, table, blockmap);
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
  } else if (kt == 1 && kh == 3 && kw == 3 && d->algo == 1) {
    // Winograd F(2x2,3x3): plan over 2x2 output blocks as a (1,4,4) stencil with stride (1,2,2)
    if (!(p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 0 && p->ph == 1 && p->pw == 1 &&
          p->dt == 1 && p->dh == 1 && p->dw == 1 && p->Hi == p->Ho && p->Wi == p->Wo &&
          p->Ti == p->To && (p->Ho % 2) == 0 && (p->Wo % 2) == 0 && d->ys_t == 0))
      return COCLR_EINVAL;
    p->Ho /= 2; p->Wo /= 2;
    p->sh = p->sw = 2;
    conv_pick_box(p, 6, 1, 4, 4);
    if (p->plane > 640) return COCLR_EINVAL;
    *variant = 60;
  } else if (kt == 1 && kh == 3 && kw == 3) {
    c = choose_tile(*p, 1, 3, 3, true, true, 256, 512);
    conv_pick_box(p, c.lbn, 1, 3, 3);
    if (c.lbn == 6) *variant = p->plane <= 256 ? 12 : 13;
    else *variant = c.bm == 128 ? 10 : 11;
  } else if (kt == 3 && kh == 1 && kw == 1 && d->algo == 2) {
    // temporal Winograd F(4,3): plan over frame QUADS as a (6,1,1) stencil with stride 4
    if (!(p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 1 && p->dt == 1 && p->dh == 1 &&
          p->dw == 1 && p->Ti == p->To &&
          (d->ys_t == 0 || (d->ys_h == 1 && d->ys_w == 1 && d->yo_h == 0 && d->yo_w == 0))))
      return COCLR_EINVAL;
    p->To = (p->To + 3) / 4;
    p->st = 4;
    conv_pick_box(p, 6, 6, 1, 1);
    if (p->plane > 384) return COCLR_EINVAL;
    *variant = 51;
  } else if (kt == 3 && kh == 1 && kw == 1 && d->algo == 1) {
    // temporal Winograd F(2,3): plan over frame PAIRS as a (4,1,1) stencil with stride 2
    if (!(p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 1 && p->dt == 1 && p->dh == 1 &&
          p->dw == 1 && p->Ti == p->To && d->ys_t == 0))
      return COCLR_EINVAL;
    p->To = (p->To + 1) / 2;
    p->st = 2;
    conv_pick_box(p, 6, 4, 1, 1);
    if (p->plane > 256) return COCLR_EINVAL;
    *variant = 50;
  } else if (kt == 3 && kh == 1 && kw == 1) {
    c = choose_tile(*p, 3, 1, 1, true, true, 256, 256);
    conv_pick_box(p, c.lbn, 3, 1, 1);
    *variant = c.lbn == 6 ? 22 : (c.bm == 128 ? 20 : 21);
  } else if (kt == 4 && kh == 1 && kw == 1 && d->algo == 2) {
    // 4-tap temporal stencil through F(2,4): plan over frame PAIRS as a (5,1,1) stencil with stride 2
    if (!(p->st == 1 && p->sh == 1 && p->sw == 1 && p->pt == 1 && p->dt == 1 && p->dh == 1 &&
          p->dw == 1 && p->Ti == p->To &&
          (d->ys_t == 0 || (d->ys_h == 1 && d->ys_w == 1 && d->yo_h == 0 && d->yo_w == 0))))
      return COCLR_EINVAL;
    p->To = (p->To + 1) / 2;
    p->st = 2;
    conv_pick_box(p, 6, 5, 1, 1);
    if (p->plane > 256) return COCLR_EINVAL;
    *variant = 52;
  } else if (kt == 4 && kh == 1 && kw == 1) {
    conv_pick_box(p, 7, 4, 1, 1);
    *variant = 25;
  } else if (kt == 1 && kh == 7 && kw == 7) {
    conv_pick_box(p, 7, 1, 7, 7);
    *variant = 30;
    if (p->Cin == 3 && p->dt == 1 && p->dh == 1 && p->dw == 1 && p->plane <= 20 * 64 &&
        p->WT <= 255 && p->WH <= 255 && p->WW <= 255) {
      *variant = 31;     // persistent stem kernel: one statistics partial per workgroup
      if (p->ntiles > kStemGrid) p->ntiles = kStemGrid;
    }
  } else if (kt == 7 && kh == 1 && kw == 1 && d->algo == 1) {
    // polyphase Winograd form of the stride-2 temporal stem conv: plan over output PAIRS as a (9,1,1)
    // stencil with stride 4
    if (!(p->st == 2 && p->sh == 1 && p->sw == 1 && p->pt == 3 && p->dt == 1 && p->dh == 1 && p->dw == 1 &&
          p->Ti == 2 * p->To && d->ys_t == 0))
      return COCLR_EINVAL;
    p->To = (p->To + 1) / 2;
    p->st = 4;
    conv_pick_box(p, 6, 9, 1, 1);
    if (p->plane > 384) return COCLR_EINVAL;
    *variant = 41;
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

extern "C" int coclr_conv3d_bwd_sums_ok(const coclr_conv_desc* d, int* ok) {
  ConvPlan p;
  int v;
  int rc = plan_forward(d, &p, &v);
  if (rc) return rc;
  *ok = (v != 60 && v != 31 && v != 51 && v != 52 && v != 41) ? 1 : 0;
  return 0;
}

namespace {

inline bool bwd_sums_variant(int variant) {
  return variant != 60 && variant != 31 && variant != 51 && variant != 52 && variant != 41;
}

// The launch behind coclr_conv3d_fwd.  With `slot`, variants that have a pair kernel fill it instead of
// launching (see PairSlot).
int conv3d_fwd_impl(const coclr_conv_desc* d, const float* x, const float* w_packed, float* y,
                    float* stats, const float* bias, const float* ep_scale, const float* ep_shift,
                    const int64_t* n_index, int relu, int accumulate, hipStream_t stream,
                    PairSlot* slot, const coclr_conv_call* bw = nullptr) {
  ConvPlan p;
  int variant;
  int rc = plan_forward(d, &p, &variant);
  if (rc) return rc;
  ConvArgs a;
  a.bwd_y = nullptr; a.bwd_scale = a.bwd_shift = a.bwd_mean = a.bwd_invstd = nullptr; a.bwd_relu = 0;
  if (bw && bw->bwd_y) {
    // backward sums of the BatchNorm unit whose dz this data gradient writes: only in the kernels whose
    // epilogue forms them (not the spatial Winograd / stem kernels), into the statistics slots, and never
    // together with an epilogue of the forward kind
    if (!bwd_sums_variant(variant) || !stats || accumulate || bias || ep_scale || relu || n_index ||
        !bw->bwd_scale || !bw->bwd_shift || !bw->bwd_mean || !bw->bwd_invstd)
      return COCLR_EINVAL;
    a.bwd_y = bw->bwd_y; a.bwd_scale = bw->bwd_scale; a.bwd_shift = bw->bwd_shift;
    a.bwd_mean = bw->bwd_mean; a.bwd_invstd = bw->bwd_invstd; a.bwd_relu = bw->bwd_relu;
  }
  a.x = x; a.w = w_packed; a.y = y; a.stats = stats; a.bias = bias;
  a.ep_scale = ep_scale; a.ep_shift = ep_shift; a.n_index = n_index;
  a.x_nstride = d->x_nstride; a.y_nstride = d->y_nstride;
  a.x_cstride = p.Ti * p.Hi * p.Wi;
  a.N = p.N; a.Cin = p.Cin; a.Cout = p.Cout;
  a.CinP = pad_to(p.Cin, 32); a.CoutP = pad_to(p.Cout, 128);
  a.Ti = p.Ti; a.Hi = p.Hi; a.Wi = p.Wi; a.To = p.To; a.Ho = p.Ho; a.Wo = p.Wo;
  a.st = p.st; a.sh = p.sh; a.sw = p.sw; a.pt = p.pt; a.ph = p.ph; a.pw = p.pw;
  a.dt = p.dt; a.dh = p.dh; a.dw = p.dw;
  // destination lattice: o*step + offset inside a (yT, yH, yW) tensor; ys_t == 0 means dense
  const bool lattice = d->ys_t > 0;
  if (lattice) {
    a.yst = d->ys_t; a.ysh = d->ys_h; a.ysw = d->ys_w;
    a.yot = d->yo_t; a.yoh = d->yo_h; a.yow = d->yo_w;
    a.yHf = d->yH; a.yWf = d->yW;
    a.y_cstride = d->yT * d->yH * d->yW;
  } else {
    a.yst = a.ysh = a.ysw = 1; a.yot = a.yoh = a.yow = 0;
    a.yHf = p.Ho; a.yWf = p.Wo;
    a.y_cstride = p.To * p.Ho * p.Wo;
  }
  a.lTW = p.lTW; a.lTH = p.lTH; a.lTT = p.lTT; a.lTN = p.lTN;
  a.nbw = p.nbw; a.nbh = p.nbh; a.nbt = p.nbt; a.nbn = p.nbn;
  a.WT = p.WT; a.WH = p.WH; a.WW = p.WW; a.plane1 = p.plane1; a.plane = p.plane;
  a.inv_plane1 = 1.0f / (float)p.plane1;
  a.inv_hw = 1.0f / (float)(p.WH * p.WW);
  a.inv_ww = 1.0f / (float)p.WW;
  a.ntiles = p.ntiles; a.mtiles = 0; a.planeS = 0; a.nchunks = 0;
  a.relu = relu; a.accumulate = accumulate;
  a.To_full = 0;
  a.in_scale = a.in_shift = nullptr; a.in_relu = 0;
  if (bw && bw->in_scale) {
    // consumer-side BatchNorm apply: only the kernel that has the operand path for it, never with a gather
    if (variant != 41 || !bw->in_shift || n_index) return COCLR_EINVAL;
    a.in_scale = bw->in_scale; a.in_shift = bw->in_shift; a.in_relu = bw->in_relu;
  }
  {
    static const bool xcd_off = getenv("COCLR_XCD_MAP") && atoi(getenv("COCLR_XCD_MAP")) == 0;
    a.xcd = !xcd_off;
  }
  // every byte offset a workgroup forms must stay below the descriptors' 2 GiB range
  const double lim = 2147483648.0;
  const double xs = n_index ? (double)(d->Nx > 0 ? d->Nx : p.N) : (double)(1 << p.lTN);
  if ((xs * (double)a.x_nstride + (double)a.Cin * a.x_cstride) * 4.0 >= lim) return COCLR_EINVAL;
  if (((double)(1 << p.lTN) * (double)a.y_nstride + (double)(a.Cout + 128) * a.y_cstride) * 4.0 >= lim)
    return COCLR_EINVAL;
  if ((double)d->kt * d->kh * d->kw * a.CinP * a.CoutP * 4.0 >= lim) return COCLR_EINVAL;
  // stencils with no reach along the flattened (H,W) axis, everything 16-byte aligned:
  // 16-byte LDS-DMA for the input window too
  const bool xv4 = d->kh == 1 && d->kw == 1 && p.Hi == 1 && p.WH == 1 && p.sw == 1 && p.lTW >= 2 &&
                   p.dt == 1 && p.dh == 1 && p.dw == 1 && (p.Wi % 4) == 0 && (p.plane % 4) == 0 &&
                   (a.x_cstride % 4) == 0 && (a.x_nstride % 4) == 0 && ((uintptr_t)x % 16) == 0;
  // (1,3,3) stride-1 pad-1 undilated stencils on rows of whole 16-byte granules: 16-byte window DMA
  static const bool xg_off = getenv("COCLR_CONV_XG") && atoi(getenv("COCLR_CONV_XG")) == 0;
  const bool xg = !xg_off && d->kt == 1 && d->kh == 3 && d->kw == 3 && p.sw == 1 && p.sh == 1 &&
                  p.st == 1 && p.pw == 1 && p.dt == 1 && p.dh == 1 && p.dw == 1 && p.lTW >= 2 &&
                  (p.Wi % 4) == 0 && (a.x_cstride % 4) == 0 && (a.x_nstride % 4) == 0 &&
                  ((uintptr_t)x % 16) == 0 && granule_count(p) <= 192;
  switch (variant) {
    case 0:  return xv4 ? launch_variant<1, 1, 1, 32, 128, 128, 2, true>(a, p, stream, slot)
                        : launch_variant<1, 1, 1, 32, 128, 128, 2>(a, p, stream);
    // 16-byte-staged kernels run with HALF the channel chunk of their 4-byte forms: the chunk is what
    // sizes the LDS stages, and two or three workgroups per CU became five or six.  Measured at B=32
    // (same box, alternating): Conv_2b 0.082 -> 0.070 ms forward / 0.070 -> 0.059 data gradient, the
    // fused heads of Mixed_3c 0.223 -> 0.20 / 0.19 -> 0.195, of Mixed_4b 0.058 -> 0.054 / 0.060 ->
    // 0.051, Conv_1a.conv2 1.36 -> 1.31; the (1,3,3) small-map kernel did not move and keeps 8.
    case 1:  return xv4 ? launch_variant<1, 1, 1, 16, 64, 128, 2, true>(a, p, stream, slot)
                        : launch_variant<1, 1, 1, 32, 64, 128, 2>(a, p, stream);
    case 2:  return xv4 ? launch_variant<1, 1, 1, 16, 64, 64, 1, true>(a, p, stream, slot)
                        : launch_variant<1, 1, 1, 32, 64, 64, 1>(a, p, stream);
    case 3:  return launch_variant<1, 1, 1, 16, 64, 64, 4>(a, p, stream);
    case 10: return xg ? launch_variant<1, 3, 3, 4, 128, 128, 3, false, true>(a, p, stream, slot)
                       : launch_variant<1, 3, 3, 4, 128, 128, 4>(a, p, stream, slot);
    case 11: return xg ? launch_variant<1, 3, 3, 8, 64, 128, 3, false, true>(a, p, stream, slot)
                       : launch_variant<1, 3, 3, 8, 64, 128, 4>(a, p, stream, slot);
    case 12: return xg ? launch_variant<1, 3, 3, 8, 64, 64, 3, false, true>(a, p, stream, slot)
                       : launch_variant<1, 3, 3, 8, 64, 64, 4>(a, p, stream, slot);
    case 13: return xg ? launch_variant<1, 3, 3, 8, 64, 64, 3, false, true>(a, p, stream, slot)
                       : launch_variant<1, 3, 3, 8, 64, 64, 8>(a, p, stream, slot);
    case 20: return xv4 ? launch_variant<3, 1, 1, 4, 128, 128, 4, true>(a, p, stream)
                        : launch_variant<3, 1, 1, 4, 128, 128, 4>(a, p, stream);
    case 21: return xv4 ? launch_variant<3, 1, 1, 8, 64, 128, 4, true>(a, p, stream)
                        : launch_variant<3, 1, 1, 8, 64, 128, 4>(a, p, stream);
    case 22: return xv4 ? launch_variant<3, 1, 1, 8, 64, 64, 4, true>(a, p, stream)
                        : launch_variant<3, 1, 1, 8, 64, 64, 4>(a, p, stream);
    case 25: return xv4 ? launch_variant<4, 1, 1, 8, 64, 128, 4, true>(a, p, stream)
                        : launch_variant<4, 1, 1, 8, 64, 128, 4>(a, p, stream);
    case 50: {
      // a.To = frame pairs; the kernel finds the frame count in yst and the plane pitch in yHf/yWf
      a.yst = d->To; a.yHf = p.Ho; a.yWf = p.Wo;
      a.y_cstride = d->To * p.Ho * p.Wo;
      a.st = 1;
      const bool xv4 = p.Hi == 1 && p.WH == 1 && p.lTW >= 2 && (p.Wi % 4) == 0 &&
                       (a.x_cstride % 4) == 0 && (a.x_nstride % 4) == 0 && ((uintptr_t)x % 16) == 0 &&
                       (p.plane % 4) == 0;
      // 8-channel chunks and a four-workgroups-per-CU register budget (accumulators in arch VGPRs,
      // 99 registers; 27 KB of LDS): four waves per SIMD instead of three.  Measured at B=32 against
      // the 16-channel / three-wave form: Conv_2c.conv2 1.02-1.06 -> 0.99 ms forward, 0.907 -> 0.874
      // data gradient; Mixed_3c.b1.conv2 0.212 -> 0.204 / 0.205 -> 0.200; the 8x8x8 layers unchanged.
      if (xv4) return launch_wino_t<8, 64, 64, 4, true, 4>(a, p, stream, slot);
      return launch_wino_t<16, 64, 64, 4, false>(a, p, stream);
    }
    case 51:
    case 52: {
      // a.To = frame quads (F(4,3)) / pairs (F(2,4)); the kernel finds the frame count in To_full and, when the
      // destination is dense, the plane pitch in yHf/yWf (a lattice along T keeps what was set above)
      a.To_full = d->To;
      if (!lattice) {
        a.yst = 1; a.yot = 0; a.yHf = p.Ho; a.yWf = p.Wo;
        a.y_cstride = d->To * p.Ho * p.Wo;
      }
      a.st = 1;
      const bool xv4 = p.Hi == 1 && p.WH == 1 && p.lTW >= 2 && (p.Wi % 4) == 0 &&
                       (a.x_cstride % 4) == 0 && (a.x_nstride % 4) == 0 && ((uintptr_t)x % 16) == 0 &&
                       (p.plane % 4) == 0;
      if (n_index) return COCLR_EINVAL;
      if (variant == 52) {
        if (xv4) return launch_wino_t24<8, 64, 64, 4, true, 3>(a, p, stream);
        return launch_wino_t24<8, 64, 64, 4, false, 3>(a, p, stream);
      }
      if (xv4) return launch_wino_t4<8, 64, 64, 6, true, 3>(a, p, stream, slot && !lattice ? slot : nullptr);
      return launch_wino_t4<8, 64, 64, 6, false, 3>(a, p, stream);
    }
    case 60: {
      // a.Ho/Wo = 2x2 blocks; the destination keeps its full row pitch
      if (n_index || ((uintptr_t)y % 8) != 0 || (a.y_nstride % 2) != 0) return COCLR_EINVAL;
      a.yHf = d->Ho; a.yWf = d->Wo;
      a.y_cstride = d->To * d->Ho * d->Wo;
      if ((((double)(1 << p.lTN)) * (double)a.y_nstride + (double)(a.Cout + 128) * a.y_cstride) * 4.0 >= lim)
        return COCLR_EINVAL;
      if (16.0 * a.CinP * a.CoutP * 4.0 >= lim) return COCLR_EINVAL;
      {
        // 16-byte window DMA when every granule of an input row is 16-byte aligned and the widened
        // window still fits three pieces per channel (LDS: 2 x (32 + 24) KiB of stages)
        static const bool x16_off = getenv("COCLR_WINO_X16") && atoi(getenv("COCLR_WINO_X16")) == 0;
        const int gran = ((p.WT * p.WH * (2 * (1 << p.lTW) + 8)) << p.lTN) / 4;
        const bool x16 = !x16_off && p.lTW >= 1 && (p.Wi % 4) == 0 && (a.x_cstride % 4) == 0 &&
                         (a.x_nstride % 4) == 0 && ((uintptr_t)x % 16) == 0 && gran <= 192;
        // two waves per SIMD (conv_wino_hw8_kernel) wherever it applies; COCLR_WINO_W8=0 (read per call:
        // an A/B and test switch) keeps the one-wave kernel
        const char* w8env = getenv("COCLR_WINO_W8");
        const bool w8 = !(w8env && w8env[0] == '0');
        if (x16 && w8) {
          int rc8 = launch_wino_hw8<8, 3>(a, p, stream);
          if (rc8 != COCLR_EINVAL) return rc8;
        }
        if (x16) {
#ifdef COCLR_WINO_ABLATE
          // timing ablations (wrong results by design; build with -DCOCLR_WINO_ABLATE, tools/wino_ablate.sh):
          // 1 no window DMA, 2 no weight DMA, 8 no statistics, 16 no patch reads + input transform,
          // 32 no weight reads.  Measured at B=32 on Conv_2c.conv1 forward (0.797 ms): 1 -> 0.737,
          // 2 -> 0.782, 3 -> 0.673, 8 -> 0.785, 16 -> 0.682, 32 -> 0.755, 48 -> 0.555.
          static const int abl = getenv("COCLR_WINO_ABL") ? atoi(getenv("COCLR_WINO_ABL")) : 0;
          switch (abl) {
            case 1: return launch_wino_hw<8, 3, true, 1>(a, p, stream);
            case 2: return launch_wino_hw<8, 3, true, 2>(a, p, stream);
            case 3: return launch_wino_hw<8, 3, true, 3>(a, p, stream);
            case 8: return launch_wino_hw<8, 3, true, 8>(a, p, stream);
            case 16: return launch_wino_hw<8, 3, true, 16>(a, p, stream);
            case 32: return launch_wino_hw<8, 3, true, 32>(a, p, stream);
            case 48: return launch_wino_hw<8, 3, true, 48>(a, p, stream);
            default: break;
          }
#endif
          return launch_wino_hw<8, 3, true>(a, p, stream);
        }
      }
      return p.plane <= 384 ? launch_wino_hw<8, 6>(a, p, stream) : launch_wino_hw<8, 10>(a, p, stream);
    }
    case 30: return launch_variant<1, 7, 7, 4, 64, 128, 20>(a, p, stream);
    case 31: return launch_stem<7, 7, 3, 20>(a, p, stream);
    case 41: {
      // a.To = output pairs; the kernel finds the frame count in yst and the plane pitch in yHf/yWf
      a.yst = d->To; a.yHf = p.Ho; a.yWf = p.Wo;
      a.y_cstride = d->To * p.Ho * p.Wo;
      a.st = 1; a.pt = 3;
      const bool xv4p = p.Hi == 1 && p.WH == 1 && p.lTW >= 2 && (p.Wi % 4) == 0 &&
                        (a.x_cstride % 4) == 0 && (a.x_nstride % 4) == 0 && ((uintptr_t)x % 16) == 0 &&
                        (p.plane % 4) == 0;
      if (n_index) return COCLR_EINVAL;
      if (a.in_scale)
        return xv4p ? launch_poly7<8, 64, 64, 6, true, 2, true>(a, p, stream)
                    : launch_poly7<8, 64, 64, 6, false, 2, true>(a, p, stream);
      if (xv4p) return launch_poly7<8, 64, 64, 6, true, 2>(a, p, stream);
      return launch_poly7<8, 64, 64, 6, false, 2>(a, p, stream);
    }
    case 40: return xv4 ? launch_variant<7, 1, 1, 4, 64, 128, 8, true>(a, p, stream)
                        : launch_variant<7, 1, 1, 8, 64, 128, 8>(a, p, stream);
  }
  return COCLR_EINVAL;
}

}  // namespace

extern "C" int coclr_conv3d_fwd(const coclr_conv_desc* d, const float* x, const float* w_packed,
                                float* y, float* stats, const float* bias, const float* ep_scale,
                                const float* ep_shift, const int64_t* n_index, int relu,
                                int accumulate, void* stream) {
  return conv3d_fwd_impl(d, x, w_packed, y, stats, bias, ep_scale, ep_shift, n_index, relu, accumulate,
                         (hipStream_t)stream, nullptr);
}

namespace {
// COCLR_PAIR=0: every problem of a multi call as its own launch, planned on its own (A/B switch, read per call)
inline bool pair_enabled() {
  const char* env = getenv("COCLR_PAIR");
  return !(env && env[0] == '0');
}
}  // namespace

extern "C" int coclr_conv3d_fwd_multi(const coclr_conv_call* calls, int n, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!calls || n < 1) return COCLR_EINVAL;
  const bool fuse = pair_enabled();
  for (int i = 0; i < n; i += 2) {
    const coclr_conv_call& c0 = calls[i];
    if (i + 1 >= n || !fuse) {
      for (int j = i; j < n && j < i + 2; ++j) {
        const coclr_conv_call& c = calls[j];
        int rc = conv3d_fwd_impl(c.d, c.x, c.w_packed, c.y, c.stats, c.bias, c.ep_scale, c.ep_shift,
                                 c.n_index, c.relu, c.accumulate, stream, nullptr, &c);
        if (rc) return rc;
      }
      continue;
    }
    const coclr_conv_call& c1 = calls[i + 1];
    PairSlot s0, s1;
    s0.pending = s1.pending = false;
    s0.pair = s1.pair = nullptr;
    s0.single = s1.single = nullptr;
    int rc = conv3d_fwd_impl(c0.d, c0.x, c0.w_packed, c0.y, c0.stats, c0.bias, c0.ep_scale, c0.ep_shift,
                             c0.n_index, c0.relu, c0.accumulate, stream, &s0, &c0);
    if (rc) return rc;
    rc = conv3d_fwd_impl(c1.d, c1.x, c1.w_packed, c1.y, c1.stats, c1.bias, c1.ep_scale, c1.ep_shift,
                         c1.n_index, c1.relu, c1.accumulate, stream, &s1, &c1);
    if (rc) return rc;
    PairFn fused = nullptr;
    if (s0.pending && s1.pending) fused = s0.single == s1.single ? s0.pair : mixed_pair(s0.single, s1.single);
    if (fused) {
      rc = fused(s0.args, s1.args, s0.blocks, s1.blocks, s0.lds > s1.lds ? s0.lds : s1.lds, stream);
      if (rc) return rc;
    } else {
      if (s0.pending) { rc = s0.single(s0.args, s0.blocks, s0.lds, stream); if (rc) return rc; }
      if (s1.pending) { rc = s1.single(s1.args, s1.blocks, s1.lds, stream); if (rc) return rc; }
    }
  }
  return 0;
}
