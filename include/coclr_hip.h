/*
 * coclr_hip.h -- C ABI of libcoclr_hip.so, the gfx950 (MI355X / CDNA4) kernel
 * library under the CoCLR training hot path.
 *
 * The reference (TengdaHan/CoCLR) has no FFI of its own: every FLOP of
 * model/pretrain.py and backbone/{s3dg,resnet_2d3d}.py is an implicit call
 * into ATen/cuDNN/NCCL.  Each entry point below therefore cites the ATen op
 * *call site* in the reference that it replaces (file:line under the
 * reference root).  The Python host (coclr_amd/) binds these with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - all tensors are device pointers to contiguous fp32 unless noted,
 *     NCDHW for activations, [Cout][Cin][kt][kh][kw] for conv weights
 *   - every function enqueues work on `stream` (a hipStream_t passed as
 *     void*; NULL = the legacy default stream), never synchronises, never
 *     allocates device memory and keeps no pointer after returning
 *   - return value: 0 on success, otherwise a hipError_t value
 *     (1 == hipErrorInvalidValue is also used for rejected arguments)
 *   - "nstride" arguments are the distance in floats between consecutive
 *     samples, so a tensor may be a channel slice of a wider buffer
 */
#ifndef COCLR_HIP_H_
#define COCLR_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------ */
/* Convolution (backbone/s3dg.py:11-13,25,39-42,59,62;                      */
/*              backbone/resnet_2d3d.py:53-59,67-79,138,192; head convs      */
/*              model/pretrain.py:52,54 when applied to un-pooled maps)      */
/* ------------------------------------------------------------------------ */

typedef struct coclr_conv_desc {
  int32_t N, Cin, Cout;
  int32_t Ti, Hi, Wi;      /* input extent  */
  int32_t To, Ho, Wo;      /* output extent */
  int32_t kt, kh, kw;      /* stencil       */
  int32_t st, sh, sw;      /* stride        */
  int32_t pt, ph, pw;      /* zero padding  */
  int32_t dt, dh, dw;      /* input dilation (zero insertion); 1 for a forward conv,
                              = forward stride when the call computes a data gradient */
  int64_t x_nstride;       /* floats between input samples  (>= Cin*Ti*Hi*Wi)  */
  int64_t y_nstride;       /* floats between output samples (>= Cout*To*Ho*Wo) */
  /* Destination lattice (phase-decomposed data gradient of a strided conv): output
   * position o of this launch is element o*ys + yo of a (yT, yH, yW) tensor.
   * ys_t == 0 selects the dense case (ys = 1, yo = 0, extent = To/Ho/Wo). */
  int32_t ys_t, ys_h, ys_w;
  int32_t yo_t, yo_h, yo_w;
  int32_t yT, yH, yW;
  int32_t Nx;              /* samples addressable through n_index (0: N) */
  int32_t algo;            /* 0: direct; 1: Winograd, w_packed must then be the transform-domain
                              operand made by coclr_conv_pack_weights(transpose | 2):
                              (3,1,1) stencil, stride 1, pad (1,0,0): F(2,3) along T, taps = 4;
                              (1,3,3) stencil, stride 1, pad (0,1,1), even Ho/Wo >= 4, dense
                              destination, no n_index: F(2x2,3x3), taps = 16;
                              (7,1,1) stencil, stride (2,1,1), pad (3,0,0), Ti = 2 To, >= 8 output
                              frames, dense destination, no n_index: polyphase form of the temporal stem
                              conv (backbone/s3dg.py:41,145) -- F(2,3) on the odd taps + F(2,4) on the even
                              ones, taps = 9 (forward operand only);
                              2: (3,1,1) stencil, stride 1, pad (1,0,0), no n_index: F(4,3) along T,
                              taps = 6 (six contractions per quad of output frames);
                              (4,1,1) stencil, stride 1, pad (1,0,0), To = Ti: F(2,4) along T, taps = 5.
                              The algo = 2 kernels also write through a destination lattice along T
                              (ys_t > 0, ys_h = ys_w = 1): the two phases of a strided temporal data gradient */
} coclr_conv_desc;

/* Number of fp32 elements of the packed-weight buffer for one conv. */
int coclr_conv_packed_size(int cin, int cout, int taps, int transpose, int64_t* elems);

/* Re-lay [Cout][Cin][taps] weights as [taps][R'][C'] (zero padded: reduction
 * channels R to x32, produced channels C to x128).
 * transpose=0: operand of the forward conv (R = Cin, C = Cout); transpose=1: operand of
 * the data gradient (R = Cout, C = Cin, stencil flipped); transpose | 2: Winograd operand
 * (coclr_conv_desc.algo >= 1) -- taps = 4: the four F(2,3) matrices of a 3-tap temporal stencil,
 * taps = 6: its six F(4,3) matrices (algo = 2), taps = 5: the five F(2,4) matrices of a 4-tap temporal stencil,
 * taps = 9: the 4 + 5 polyphase matrices of the 7-tap stride-2 stem conv (the source taps are then tap_base +
 * t*tap_step, t = 0..6; transpose = 0 only),
 * taps = 16: the sixteen F(2x2,3x3) matrices U = G g G^T of a 9-tap spatial stencil, laid out
 * [R'][C'][16] (stand-alone operands only).  co/ci strides, tap_base and
 * tap_step address a sub-stencil: source tap of packed tap t is tap_base + t*tap_step
 * (one kt-slice of a (5,7,7) stem; the taps of one phase of a strided data gradient).
 * rows_total/cols_total > 0 place this tensor at (row0, col0) of a WIDER packed operand
 * [taps][pad32(rows_total)][pad128(cols_total)] that the caller has zero-filled: several
 * convs that read the same input (the three 1x1x1 heads of an inception block,
 * backbone/s3dg.py:97-112) then run as one convolution over concatenated channels. */
int coclr_conv_pack_weights(const float* w, float* packed, int cout, int cin, int taps,
                            int64_t co_stride, int64_t ci_stride, int tap_base, int tap_step,
                            int transpose, int row0, int rows_total, int col0, int cols_total,
                            void* stream);

/* The same re-layouts, many at once (every operand of an encoder: ~80 tiny launches per pass
 * otherwise).  coclr_conv_pack_describe validates one request exactly as
 * coclr_conv_pack_weights would and writes it as a 16-word table row plus the number of
 * 1024-element blocks it needs; the caller uploads the rows and a block map
 * int32[nblocks][2] = {row, block index within the row} and launches them all with
 * coclr_conv_pack_batch.  The pointers in a row must stay valid for every replay. */
int coclr_conv_pack_describe(const float* w, float* packed, int cout, int cin, int taps,
                             int64_t co_stride, int64_t ci_stride, int tap_base, int tap_step,
                             int transpose, int row0, int rows_total, int col0, int cols_total,
                             int64_t* entry, int32_t* nblocks);
int coclr_conv_pack_batch(const int64_t* table, const int32_t* blockmap, int nblocks,
                          void* stream);

/* Number of per-workgroup BatchNorm partial sums coclr_conv3d_fwd will emit
 * per channel for this geometry (stats buffer = 2 * Cout * ntiles floats). */
int coclr_conv3d_ntiles(const coclr_conv_desc* d, int* ntiles);

/* y (+)= act(affine(conv(x, w) + bias)).  Replaces aten::conv3d forward and,
 * with transposed weights + input dilation, the dgrad half of
 * aten::convolution_backward.  stats (optional): [2][Cout][ntiles] partial
 * sum / sum-of-squares of the raw conv output for train-mode BatchNorm
 * (backbone/s3dg.py:16,46-47).  n_index (optional): x sample n is read from
 * sample n_index[n] -- the shuffle-BN row gather of model/pretrain.py:124
 * folded into the first conv. */
int coclr_conv3d_fwd(const coclr_conv_desc* d, const float* x, const float* w_packed, float* y,
                     float* stats, const float* bias, const float* ep_scale,
                     const float* ep_shift, const int64_t* n_index, int relu, int accumulate,
                     void* stream);

/* Several INDEPENDENT convolutions in one call; consecutive pairs (0,1), (2,3), ... that resolve to the
 * same kernel variant run as ONE launch whose grid holds the tiles of both problems.  Replaces two
 * aten::conv3d calls of sibling branches: the (1,3,3) / (3,1,1) convolutions of branch1 and branch2 of an
 * inception block (backbone/s3dg.py:100-118), forward and data gradient -- on the 8x8x8 / 4x4x4 maps each
 * of them alone is a 10-40 us launch that leaves most of the chip idle.  Per-problem arguments are those
 * of coclr_conv3d_fwd.  Every problem keeps the plan it would have alone (same tiles, same statistics
 * layout: coclr_conv3d_ntiles), so outputs and statistics are bit-identical to separate calls.
 * COCLR_PAIR=0 in the environment launches every problem on its own. */
typedef struct coclr_conv_call {
  const coclr_conv_desc* d;
  const float* x;
  const float* w_packed;
  float* y;
  float* stats;
  const float* bias;
  const float* ep_scale;
  const float* ep_shift;
  const int64_t* n_index;
  int32_t relu, accumulate;
  /* Data gradient whose destination y is the dz of a BatchNorm(+ReLU) unit (the producer of the tensor
   * this convolution read in forward): bwd_y = that unit's convolution output, laid out exactly like y,
   * bwd_scale / bwd_shift / bwd_mean / bwd_invstd its per-channel coefficients.  The kernel then also
   * forms the unit's backward sums while dz is in registers -- g = dz where bwd_relu == 0 or
   * bwd_y*scale + shift > 0; stats[0][c][tile] = sum g, stats[1][c][tile] = sum g*(bwd_y - mean)*invstd --
   * and coclr_bn_act_backward_multi takes them as `part` instead of running its reduction pass over dz
   * and y (aten::native_batch_norm_backward's sums + threshold_backward; backbone/s3dg.py:46-48,60-64).
   * Requires stats, no accumulate / bias / ep_* / relu / n_index, and a kernel that has the epilogue
   * (coclr_conv3d_bwd_sums_ok; COCLR_EINVAL otherwise).  All NULL / 0: a plain call. */
  const float* bwd_y;
  const float* bwd_scale;
  const float* bwd_shift;
  const float* bwd_mean;
  const float* bwd_invstd;
  int32_t bwd_relu;
  /* The input x is the RAW convolution output of a BatchNorm(+ReLU) unit whose apply pass has not run:
   * the kernel applies x' = x * in_scale[ci] + in_shift[ci] (max(., 0) when in_relu) to every element it
   * reads, zero padding staying zero -- the normalised tensor is never written or re-read (the apply pass
   * of backbone/s3dg.py:46-48 fused into the NEXT convolution of the separable unit, :49).  Only the
   * kernels that move their B operand through registers support it (the polyphase temporal stem conv);
   * COCLR_EINVAL otherwise.  NULL: a plain call. */
  int32_t in_relu;
  const float* in_scale;
  const float* in_shift;
} coclr_conv_call;
int coclr_conv3d_fwd_multi(const coclr_conv_call* calls, int n, void* stream);
int coclr_conv3d_bwd_sums_ok(const coclr_conv_desc* d, int* ok);

/* Split-K workspace (fp32 elements) for coclr_conv3d_wgrad. */
int coclr_conv3d_wgrad_workspace(const coclr_conv_desc* d, int64_t* elems);

/* dw[co][ci][tap] (+)= sum_{n,o} dy[n][co][o] * x[n][ci][o*s - p + tap]: the wgrad
 * half of aten::convolution_backward.  dw is addressed as
 * dw[co*w_co_stride + ci*w_ci_stride + tap_base + tap]. */
int coclr_conv3d_wgrad(const coclr_conv_desc* d, const float* x, const float* dy, float* dw,
                       float* workspace, int64_t w_co_stride, int64_t w_ci_stride, int tap_base,
                       int accumulate, void* stream);

/* The weight gradient of a convolution whose BatchNorm(+ReLU) backward apply pass has NOT run: `dz` is
 * d(activation) of the unit behind the convolution (backbone/s3dg.py:24-28: conv -> bn -> relu), `y` the
 * convolution's own output, coef[5][Cout] what coclr_bn_act_backward_coeffs left.  The kernel forms
 *   dy = A[co] * g + B[co] * y + D[co],   g = relu ? (y * scale[co] + shift[co] > 0 ? dz : 0) : dz
 * -- the expression of coclr_bn_act_backward's second pass, bit for bit -- between LDS and the matrix
 * pipe, so d(conv output) is never written or re-read.  For units whose data gradient nobody needs (the
 * (1,7,7) stem over the clip: 1 GB per pass at B = 32); coclr_conv3d_wgrad_bn_ok says whether the geometry
 * has such a kernel, COCLR_EINVAL otherwise.  Replaces the native_batch_norm_backward +
 * convolution_backward pair autograd runs for backbone/s3dg.py:145 (Conv_1a.conv1 / bn1). */
int coclr_conv3d_wgrad_bn_ok(const coclr_conv_desc* d, int* ok);
int coclr_conv3d_wgrad_bn(const coclr_conv_desc* d, const float* x, const float* dz, const float* y,
                          int64_t y_nstride, const float* coef, int relu, float* dw, float* workspace,
                          int64_t w_co_stride, int64_t w_ci_stride, int tap_base, int accumulate,
                          void* stream);

/* The same gradient delivered to up to four destinations: output-channel rows
 * [row_end[i-1], row_end[i]) go to dw_list[i] (row index local to the destination;
 * row_end[nseg-1] == d->Cout).  For convolutions that stand for several parameters at once --
 * the three 1x1x1 heads of an inception block run as one convolution over concatenated output
 * channels (backbone/s3dg.py:97-104,119-123) -- whose gradients live at unrelated addresses
 * (views of DistributedDataParallel's buckets, main_nce.py:172).  dw_list / row_end are HOST
 * arrays read during the call. */
int coclr_conv3d_wgrad_multi(const coclr_conv_desc* d, const float* x, const float* dy,
                             float* const* dw_list, const int32_t* row_end, int nseg,
                             float* workspace, int64_t w_co_stride, int64_t w_ci_stride,
                             int tap_base, int accumulate, void* stream);

/* ------------------------------------------------------------------------ */
/* BatchNorm3d + ReLU (+ residual)  (backbone/s3dg.py:16-17,26-27,46-48,     */
/*                                   60-64; backbone/resnet_2d3d.py:54-83)   */
/* ------------------------------------------------------------------------ */

/* Fold the conv partial sums (sum[c][ntiles], sumsq[c][ntiles]; two pointers so a channel
 * range of a concatenated convolution can be finalised on its own): batch mean / invstd,
 * fused scale = gamma*invstd and shift = beta - mean*scale; momentum update of
 * running_mean / running_var (unbiased) and num_batches_tracked += 1 (aten::batch_norm,
 * training=True). */
int coclr_bn_finalize(const float* sum, const float* sumsq, int C, int ntiles, double count,
                      const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                      float* mean, float* invstd, float* scale, float* shift, void* stream);

/* coclr_bn_finalize followed by coclr_bn_act_apply (no residual) as ONE call: for layers whose
 * channels hold few values (N*S <= 32768: the 8x8x8 and 4x4x4 maps of S3D's last two stages) both
 * run in a single launch, one workgroup per channel; larger layers take the two launches. */
int coclr_bn_finalize_apply(const float* sum, const float* sumsq, int C, int ntiles, double count,
                            const float* gamma, const float* beta, float* running_mean,
                            float* running_var, int64_t* num_batches_tracked, float momentum,
                            float eps, float* mean, float* invstd, float* scale, float* shift,
                            const float* y, float* z, int N, int64_t S, int64_t y_nstride,
                            int64_t z_nstride, int relu, void* stream);

/* Eval-mode coefficients from the running statistics (main_coclr.py:363). */
int coclr_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* mean, float* invstd,
                         float* scale, float* shift, void* stream);

/* z = act(y*scale[c] + shift[c] (+ residual)); every tensor is [N][C][S] with its own
 * sample stride (channel slices of wider buffers). */
int coclr_bn_act_apply(const float* y, const float* scale, const float* shift,
                       const float* residual, float* z, int N, int C, int64_t S, int64_t y_nstride,
                       int64_t z_nstride, int64_t res_nstride, int relu, void* stream);

/* Backward of the above (aten::threshold_backward + native_batch_norm_backward):
 * dy, dgamma, dbeta and, for residual units, dres (+)= masked dz.
 * z may be NULL (ReLU mask is then recomputed from y).  sums_ws: fp64 partial sums, one
 * pair per (channel, sample) -- coclr_bn_backward_workspace doubles; no zero-fill needed,
 * no atomics (run-to-run deterministic). */
int coclr_bn_backward_workspace(int N, int C, int64_t* doubles);
int coclr_bn_act_backward(const float* dz, const float* y, const float* z, const float* scale,
                          const float* shift, const float* mean, const float* invstd,
                          double* sums_ws, float* dy, float* dres, float* dgamma, float* dbeta,
                          int N, int C, int64_t S, int64_t dz_nstride, int64_t y_nstride,
                          int64_t dy_nstride, int64_t z_nstride, int64_t dres_nstride, int relu,
                          int training, int dres_accumulate, void* stream);

/* The first pass of the above plus the coefficients of its second, for a unit whose d(conv output) is
 * applied by its one reader (coclr_conv3d_wgrad_bn): coef[5][C] = A, B, D, scale, shift with
 * dy = A*g + B*y + D (training: A = scale, B = -scale*invstd*mean(g*xhat),
 * D = scale*(mean*invstd*mean(g*xhat) - mean(g)); frozen statistics: dy = scale*g); dgamma / dbeta as
 * above.  Two launches, no pass that writes dy. */
int coclr_bn_act_backward_coeffs(const float* dz, const float* y, const float* scale, const float* shift,
                                 const float* mean, const float* invstd, double* sums_ws, float* coef,
                                 float* dgamma, float* dbeta, int N, int C, int64_t S,
                                 int64_t dz_nstride, int64_t y_nstride, int relu, int training,
                                 void* stream);

/* ------------------------------------------------------------------------ */
/* Pooling (backbone/s3dg.py:105,151,162,173,190; resnet_2d3d.py:141;        */
/*          model/pretrain.py:51)                                            */
/* ------------------------------------------------------------------------ */

typedef struct coclr_pool_desc {
  int32_t N, C;
  int32_t Ti, Hi, Wi, To, Ho, Wo;
  int32_t kt, kh, kw, st, sh, sw, pt, ph, pw;
  int64_t x_nstride, y_nstride;
} coclr_pool_desc;

/* aten::max_pool3d_with_indices (floor mode, -inf padding, first max wins);
 * indices (optional, [N][C][To*Ho*Wo] int32) = flat offset inside the input plane.
 * in_scale / in_shift (optional, [C]): the input is read as x*in_scale[c] + in_shift[c], clamped at 0
 * when in_relu -- the BatchNorm3d + ReLU in front of the pool (backbone/s3dg.py:60-64 before
 * :151,:162) applied on the fly, so that the normalised activation is never written to HBM. */
int coclr_maxpool3d_fwd(const coclr_pool_desc* d, const float* x, float* y, int32_t* indices,
                        const float* in_scale, const float* in_shift, int in_relu, void* stream);
/* aten::max_pool3d_with_indices_backward, gather form (deterministic). */
int coclr_maxpool3d_bwd(const coclr_pool_desc* d, const float* dy, const int32_t* indices, float* dx,
                        int64_t dy_nstride, int64_t dx_nstride, int accumulate, void* stream);

/* Several BatchNorm units in one call: runs of up to four units that are small enough for the
 * one-workgroup-per-channel form (N*S <= 32768, the last two stages of S3D) and share a vector width run
 * as ONE launch whose grid is their channels back to back; everything else goes through the single-unit
 * entry points, in order.  Fields as the arguments of coclr_bn_finalize_apply / coclr_bn_act_backward
 * (no residual).  The three 1x1x1 heads of an inception block and the two separable branches side by
 * side (backbone/s3dg.py:97-118): 10 us launches of 16-384 workgroups each. */
typedef struct coclr_bn_fwd_call {
  const float* sum; const float* sumsq; const float* gamma; const float* beta;
  float* running_mean; float* running_var; int64_t* num_batches_tracked;
  float* mean; float* invstd; float* scale; float* shift;
  const float* y; float* z;
  double count;
  int64_t S, y_nstride, z_nstride;
  int32_t C, ntiles, N, relu;
  float momentum, eps;
} coclr_bn_fwd_call;
typedef struct coclr_bn_bwd_call {
  const float* dz; const float* y; const float* scale; const float* shift; const float* mean;
  const float* invstd;
  double* sums_ws; float* dy; float* dgamma; float* dbeta;
  int64_t S, dz_nstride, y_nstride, dy_nstride;
  int32_t N, C, relu, training;
  /* Backward sums already formed by the data gradient(s) that wrote dz (coclr_conv_call.bwd_y): up to two
   * [2][C][part_ntiles[i]] arrays (a strided convolution's data gradient is one launch per residue class).
   * part[0] != NULL: the reduction pass is replaced by a fold of these partials (fp64, fixed order). */
  const float* part[2];
  int32_t part_ntiles[2];
} coclr_bn_bwd_call;
int coclr_bn_finalize_apply_multi(const coclr_bn_fwd_call* calls, int n, void* stream);
int coclr_bn_act_backward_multi(const coclr_bn_bwd_call* calls, int n, void* stream);

/* BatchNorm(+ReLU) backward of a unit whose only consumer is the max-pool described by `d` that
 * applied the unit's affine + ReLU while reading (coclr_maxpool3d_fwd with in_scale / in_shift): the
 * gradient of the normalised activation is the pool's dy at the arg-max positions and is never
 * written.  Replaces aten::max_pool3d_with_indices_backward + aten::native_batch_norm_backward +
 * threshold_backward (backbone/s3dg.py:151,162 behind :60-64).  y: the unit's convolution output
 * [N][C][Ti][Hi][Wi]; pool_dy / pool_idx: [N][C][To][Ho][Wo]; sums: workspace of
 * coclr_bn_backward_workspace(N, C) doubles.  Only for pools whose (time-folded) input plane fits the
 * kernel's LDS tile: coclr_bn_act_backward_pooled_fits answers that from the same tiling constants;
 * a pool that does not fit is rejected with COCLR_EINVAL (1) and the caller runs
 * coclr_maxpool3d_bwd + coclr_bn_act_backward instead. */
int coclr_bn_act_backward_pooled(const coclr_pool_desc* d, const float* pool_dy,
                                 const int32_t* pool_idx, const float* y, const float* scale,
                                 const float* shift, const float* mean, const float* invstd,
                                 double* sums, float* dy, float* dgamma, float* dbeta,
                                 int64_t pool_dy_nstride, int64_t y_nstride, int64_t dy_nstride,
                                 int relu, int training, void* stream);
int coclr_bn_act_backward_pooled_fits(const coclr_pool_desc* d, int* fits);
/* aten::adaptive_avg_pool3d(x, (1,1,1)) and its backward; planes = N*C. */
int coclr_global_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t S, void* stream);
int coclr_global_avgpool_bwd(const float* dy, float* dx, int64_t planes, int64_t S, void* stream);

/* ------------------------------------------------------------------------ */
/* Contrastive head (model/pretrain.py)                                      */
/* ------------------------------------------------------------------------ */

/* C[m][n] (+)= act(alpha * sum_k A(m,k)*B(k,n) + bias[n]) on the fp32 MFMA;
 * A(m,k) = a[m*sam + k*sak], B(k,n) = b[k*sbk + n*sbn].  splits > 1 selects
 * split-K through `workspace` (coclr_gemm_workspace elements).  Used for the
 * projection-head 1x1x1 convs on pooled features (pretrain.py:52,54), their
 * backward, and the similarity products below. */
int coclr_gemm_workspace(int M, int N, int K, int splits, int64_t* elems);
int coclr_gemm(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk, int64_t sbn,
               float* c, int64_t ldc, const float* bias, int M, int N, int K, float alpha, int relu,
               int accumulate, int splits, float* workspace, void* stream);

/* The products of an encoder's projection head and of its backward, each with the row-level operation
 * that FOLLOWS it in the reference applied by the kernel that folds the split-K partials
 * (pretrain.py:49-54: AdaptiveAvgPool3d -> Conv3d 1x1x1 -> ReLU -> Conv3d 1x1x1; :153-154 / :166-167
 * F.normalize; :175-182 the logits whose gradient arrives here).  Two launches per product instead of three
 * or four, same fold order as coclr_gemm (k = 0 .. splits-1: bit-identical to the unfused sequence):
 *   mode 0  none (with splits == 1: the plain product in one launch, optionally + rowsum)
 *   mode 1  aten::threshold_backward: c = a[m][n] > 0 ? v : 0            (nn.ReLU backward, pretrain.py:53)
 *   mode 2  F.normalize rows: c = v / max(||v||, f), out2[m] = 1 / max(||v||, f)              (N <= 512)
 *   mode 3  v += a[m*lda] * f * b[m][n]  (the l_pos term of pretrain.py:175, f = 1/T), then the backward of
 *           F.normalize: c = (v - y <y, v>) * inv_norm[m]                                      (N <= 512)
 *   mode 4  aten::adaptive_avg_pool3d backward: c is dense [M][N][S], c[m][n][:] = v / S
 * rowsum (splits == 1): rowsum[m] = sum_k A(m,k) -- the bias gradient of a weight-gradient product.
 * workspace: max(splits, 1) * M * N elements (always used unless mode 0 with splits == 1). */
typedef struct coclr_gemm_epilogue {
  int32_t mode;
  int32_t S;
  const float* a;
  int64_t lda;
  const float* b;
  const float* y;
  const float* inv_norm;
  float* out2;
  float f;
  float* rowsum;
} coclr_gemm_epilogue;
int coclr_gemm_fused(const float* a, int64_t sam, int64_t sak, const float* b, int64_t sbk, int64_t sbn,
                     float* c, int64_t ldc, const float* bias, int M, int N, int K, float alpha, int relu,
                     int splits, float* workspace, const coclr_gemm_epilogue* ep, void* stream);

/* F.normalize(x, dim=1) over rows of D (pretrain.py:154,167,380) and backward. */
int coclr_l2norm_fwd(const float* x, float* y, float* inv_norm, int rows, int D, float eps,
                     void* stream);
int coclr_l2norm_bwd(const float* dy, const float* y, const float* inv_norm, float* dx, int rows,
                     int D, void* stream);

/* logits[B][1+K] = [ <q_b,k_b> | q . queue ] / T  (pretrain.py:175-182: two einsums,
 * queue.clone(), cat and the in-place divide in one pass). queue is [D][K]. */
int coclr_nce_logits_fwd(const float* q, const float* k, const float* queue, float* logits, int B,
                         int D, int K, float T, void* stream);
/* dq = (dlogits[:,1:] . queue^T + dlogits[:,0] * k) / T ; workspace as coclr_gemm(B, D, K, splits). */
int coclr_nce_logits_bwd(const float* dlogits, const float* k, const float* queue, float* dq,
                         float* workspace, int B, int D, int K, float T, int splits, void* stream);

/* p_k = p_k*m + p_q*(1-m) over many tensors in one launch (pretrain.py:76-80).
 * table: int64[3*nchunks] on device = {dst ptr, src ptr, count<=65536} per chunk. */
int coclr_momentum_update(const int64_t* table, int nchunks, float m, float one_minus_m,
                          void* stream);

/* queue[:, ptr:ptr+BW] = keys^T with ptr read on device (pretrain.py:89-93 without the
 * int(queue_ptr) host sync); int64 side queues (pretrain.py:217,337-338); ptr=(ptr+BW)%K. */
int coclr_queue_enqueue(float* queue, const float* keys, int D, int K, int BW, const int64_t* ptr,
                        void* stream);
int coclr_queue_fill_i64(int64_t* queue, const int64_t* vals, int64_t const_val, int K, int BW,
                         const int64_t* ptr, void* stream);
int coclr_queue_advance(int64_t* ptr, int BW, int K, void* stream);

/* mask[B][1+K] (bytes): col 0 = 1; col 1+j = (src[b]==names[j]) or j among the topk
 * largest sim[b][:] after same-source columns are set to -inf
 * (pretrain.py:397-413; with topk=0 also UberNCE's label mask, pretrain.py:267-269). */
int coclr_positive_mask(const float* sim, const int64_t* src, const int64_t* names, uint8_t* mask,
                        int B, int K, int topk, void* stream);

/* The same mask with the similarity product folded in and no (B, K) similarity tensor
 * (pretrain.py:405-410: `sim = kf.matmul(queue_second)`, siblings to -inf, topk, scatter): ONE launch,
 * sim tiles on the MFMA pipe, a running top-k per row across tiles.  kf [B][D] unit rows, queue_second
 * [D][K]; D must be 128, topk <= 16.  Workspaces: cand_val / cand_idx [B][ceil(K/64)][topk];
 * counters [ceil(B/32)] int32, ZERO before the first call (every call leaves them zero again).
 * sim_out (optional, [B][K]) receives the similarities, for tests. */
int coclr_mine_positives(const float* kf, const float* queue_second, const int64_t* src,
                         const int64_t* names, uint8_t* mask, float* cand_val, int32_t* cand_idx,
                         int32_t* counters, float* sim_out, int B, int D, int K, int topk,
                         void* stream);

/* out[i][:] = in[idx[i]][:] (pretrain.py:124,143); rows of `row_elems` floats, source rows
 * `in_row_stride` floats apart (>= row_elems: the second clip of a (B,2,...) pair is a
 * strided view). */
int coclr_gather_rows(const float* in, const int64_t* idx, float* out, int rows, int64_t row_elems,
                      int64_t in_row_stride, void* stream);

/* out[i][:] = the `row_elems` floats at device address row_ptrs[i] (int64[rows] on the device): the
 * shuffle-BN exchange of pretrain.py:98-124 as a ROW PULL -- each rank reads the B key clips it will
 * encode straight out of its peers' staging buffers (hipIpc-mapped, over xGMI) instead of all-gathering
 * B*world clips; local rows are ordinary addresses. */
int coclr_pull_rows(const int64_t* row_ptrs, float* out, int rows, int64_t row_elems, void* stream);

/* nn.ReLU of the projection head (pretrain.py:53) and small helpers. */
int coclr_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int coclr_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
int coclr_colsum(const float* x, float* out, int rows, int cols, void* stream);

/* S3D-G self gating (backbone/s3dg.py:68-78): out = x * sigmoid(fc(mean_{T,H,W} x)).
 * The mean is coclr_global_avgpool_fwd, fc is coclr_gemm; these supply the rest:
 * w = sigmoid(s) and ds = dw*w*(1-w); out[n][c][:] (+)= a[n][c][:]*gain[n*C+c] + bias[n*C+c]
 * (forward scaling, and dx = dout*w + (ds.W)/S in the backward); out[n*C+c] = <a, b> per
 * (n, c) plane (dw of the backward). */
int coclr_sigmoid_fwd(const float* s, float* w, int64_t n, void* stream);
int coclr_sigmoid_bwd(const float* dw, const float* w, float* ds, int64_t n, void* stream);
int coclr_plane_scale(const float* a, const float* gain, const float* bias, float* out, int N, int C,
                      int64_t S, int64_t a_nstride, int64_t out_nstride, int accumulate,
                      void* stream);
int coclr_plane_dot(const float* a, const float* b, float* out, int N, int C, int64_t S,
                    int64_t a_nstride, int64_t b_nstride, void* stream);

/* ------------------------------------------------------------------------ */
/* Training-loop neighbours of the model: optimiser step,     */
/* loss + accuracy epilogue, input staging                                   */
/* ------------------------------------------------------------------------ */

/* torch.optim.Adam.step() over the launch scripts' one-group-per-tensor parameter list
 * (main_nce.py:190-200,331; main_coclr.py:213,406) as ONE launch, operation-for-operation the
 * arithmetic of torch.optim.Adam (amsgrad=False, maximize=False, L2 weight decay).
 *   table  int64[nchunks][8] on device: {param, grad, exp_avg, exp_avg_sq, key_param or 0,
 *          count (<= 65536 elements of the tensor), group index, 0}
 *   hyper  double[.][8] on device, one row per group: {lr, beta1, beta2, eps, weight_decay, 0,0,0}
 *          (doubles: torch forms 1-beta and beta**t from Python floats)
 *   steps  float[.] on device, one per group: number of steps taken so far; read as t = steps+1
 *          for the bias corrections, then incremented for the `ngroups` groups listed in `groups`
 * key_param != 0 folds the momentum-encoder update of model/pretrain.py:76-80 into the pass:
 *   key = key*mom_m + param_new*mom_1m  (exactly what the next forward would compute). */
int coclr_adam_step(const int64_t* table, int nchunks, const double* hyper, float* steps,
                    const int32_t* groups, int ngroups, float mom_m, float mom_1m, void* stream);

/* Loss + accuracy over logits[B][N1] in one pass, results as device scalars:
 *   mode 0  nn.CrossEntropyLoss(logits, target)                         (main_nce.py:315)
 *   mode 1  multi_nce_loss: -log(sum_j softmax_j*mask_j)                (main_coclr.py:343-346);
 *           drop_self: column 0 is left out of a row's positives when the row has others
 *           (the mask_clone[mask_sum!=1, 0] = 0 branch, main_coclr.py:384-389)
 *   mode 2  -(sum_j log_softmax_j*mask_j) / sum_j mask_j                (main_nce.py:322)
 * scalars[5] = batch means of {loss, hit@k1, hit@k2, self-hit@k1, self-hit@k2}: hit@k = one of the
 * row's positives is among its k largest logits (calc_topk_accuracy / calc_mask_accuracy,
 * utils/utils.py:52-85), self-hit@k the same for column 0 alone (main_coclr.py:392).
 * rowstats float[B][8] and flags uint8[B] carry what the backward needs.  mask: bytes, [B][N1]. */
int coclr_nce_loss_fwd(const float* logits, const uint8_t* mask, const int64_t* target,
                       float* rowstats, uint8_t* flags, float* scalars, int B, int N1, int mode,
                       int drop_self, int k1, int k2, void* stream);
/* dlogits = dloss[0]/B * (softmax - positives' weights); dloss is a DEVICE scalar. */
int coclr_nce_loss_bwd(const float* logits, const uint8_t* mask, const int64_t* target,
                       const float* rowstats, const uint8_t* flags, const float* dloss,
                       float* dlogits, int B, int N1, int mode, void* stream);

/* Loader frames -> model input in one pass (main_nce.py:207-209,299-302; utils/transforms.py:57-63;
 * model/pretrain.py:149-150): frames[B][C][S][THW] (uint8 when from_u8, else fp32 in [0,1]) ->
 * out[B][S][C][THW] fp32 = ((x / 255 if uint8) - mean[c]) / std[c].  THW = seq_len*H*W; mean / std
 * are HOST arrays of C floats read at call time. */
int coclr_stage_clips(const void* frames, int from_u8, float* out, int B, int C, int S, int64_t THW,
                      const float* mean, const float* std, void* stream);

/* ------------------------------------------------------------------------ */
/* Evaluation consumers (model/classifier.py:47-61; eval/main_classifier.py) */
/* ------------------------------------------------------------------------ */

/* fp32 workspace elements for the two column-statistics entry points below. */
int coclr_colstats_workspace(int rows, int cols, int64_t* elems);
/* BatchNorm1d batch statistics of pooled features x[rows][cols] (final_bn, classifier.py:34,56)
 * as stats[2][cols] = {sum, sum of squares}: the layout coclr_bn_finalize takes with ntiles=1. */
int coclr_bn1d_stats(const float* x, float* stats, float* workspace, int rows, int cols,
                     void* stream);
/* out = x - x.mean(0) (eval/main_classifier.py:690-691). */
int coclr_center_rows(const float* x, float* out, float* workspace, int rows, int cols,
                      void* stream);
/* Nearest-neighbour retrieval (eval/main_classifier.py:699-703): for every row of sim[B][N] the
 * kmax most similar columns in order (value descending, column ascending on ties);
 * hits[b][i] = 1.0 if any(train_label[top ks[i] columns] == test_label[b]) else 0.0 (fp32, so
 * coclr_colsum gives the accuracies); ks ascending, ks[nk-1] ==
 * kmax.  topidx (optional) int32[B][kmax] receives the selected columns. */
int coclr_retrieval_hits(const float* sim, const int64_t* train_label, const int64_t* test_label,
                         const int32_t* ks, int nk, float* hits, int32_t* topidx, int B, int N,
                         int kmax, void* stream);

/* Library/ABI version, bumped when a signature changes. */
int coclr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* COCLR_HIP_H_ */
