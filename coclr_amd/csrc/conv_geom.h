// Host-side geometry planning shared by the forward/dgrad and wgrad conv
// launchers: picks the output "box" (a power-of-two brick of output positions
// owned by one workgroup) and derives the input stencil window that gets
// staged in LDS.
#pragma once
#include "common.h"
#include "../../include/coclr_hip.h"

struct ConvPlan {
  // normalised geometry (spatial dims may be flattened for kernels with no
  // extent/stride/pad on that axis)
  int N, Cin, Cout;
  int Ti, Hi, Wi, To, Ho, Wo;
  int st, sh, sw, pt, ph, pw, dt, dh, dw;
  // box: log2 extents (w, h, t, n), number of boxes per axis
  int lTW, lTH, lTT, lTN;
  int nbw, nbh, nbt, nbn;
  int ntiles;   // partial-statistics slots a forward launch emits per channel
  int nboxes;   // boxes covering the output
  // per-sample window extents and sizes
  int WT, WH, WW, plane1, plane;
};

static inline int ceil_log2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

// Collapse axes the stencil does not touch so boxes become contiguous runs.
static inline void conv_normalise(const coclr_conv_desc* d, ConvPlan* p) {
  p->N = d->N; p->Cin = d->Cin; p->Cout = d->Cout;
  p->Ti = d->Ti; p->Hi = d->Hi; p->Wi = d->Wi;
  p->To = d->To; p->Ho = d->Ho; p->Wo = d->Wo;
  p->st = d->st; p->sh = d->sh; p->sw = d->sw;
  p->pt = d->pt; p->ph = d->ph; p->pw = d->pw;
  p->dt = d->dt; p->dh = d->dh; p->dw = d->dw;
  // a destination lattice (ys_t > 0) pins the axes it steps along
  const bool lat = d->ys_t > 0;
  const bool h_free = d->kh == 1 && d->sh == 1 && d->ph == 0 && d->dh == 1 && d->Hi == d->Ho &&
                      (!lat || (d->ys_h == 1 && d->yo_h == 0 && d->yH == d->Ho));
  const bool w_free = d->kw == 1 && d->sw == 1 && d->pw == 0 && d->dw == 1 && d->Wi == d->Wo &&
                      (!lat || (d->ys_w == 1 && d->yo_w == 0 && d->yW == d->Wo));
  const bool t_free = d->kt == 1 && d->st == 1 && d->pt == 0 && d->dt == 1 && d->Ti == d->To &&
                      (!lat || (d->ys_t == 1 && d->yo_t == 0 && d->yT == d->To));
  if (h_free && w_free) {
    p->Wi = p->Wo = d->Hi * d->Wi;
    p->Hi = p->Ho = 1;
    if (t_free) {
      p->Wi = p->Wo = p->Wi * d->Ti;
      p->Ti = p->To = 1;
    }
  }
}

// Choose the box for a BN = 2^lbn position tile.  `kt/kh/kw` are the stencil
// extents of the launch (after any slicing).
static inline void conv_pick_box(ConvPlan* p, int lbn, int kt, int kh, int kw) {
  const bool spatial = (kh > 1) || (kw > 1) || p->sh > 1 || p->sw > 1;
  int capw = spatial ? (p->sw > 1 ? 6 : 5) : (kt > 1 || p->st > 1 ? 4 : 7);
  int lw = ceil_log2(p->Wo); if (lw > capw) lw = capw; if (lw > lbn) lw = lbn;
  int rest = lbn - lw;
  int lh = ceil_log2(p->Ho); if (lh > rest) lh = rest;
  rest -= lh;
  int lt = ceil_log2(p->To); if (lt > rest) lt = rest;
  rest -= lt;
  int ln = rest;  // whatever is left spans samples
  // do not span more samples than exist (keeps masked work bounded)
  p->lTW = lw; p->lTH = lh; p->lTT = lt; p->lTN = ln;
  p->nbw = cdiv(p->Wo, 1 << lw);
  p->nbh = cdiv(p->Ho, 1 << lh);
  p->nbt = cdiv(p->To, 1 << lt);
  p->nbn = cdiv(p->N, 1 << ln);
  p->ntiles = p->nbw * p->nbh * p->nbt * p->nbn;
  p->nboxes = p->ntiles;
  p->WT = ((1 << lt) - 1) * p->st + kt;
  p->WH = ((1 << lh) - 1) * p->sh + kh;
  p->WW = ((1 << lw) - 1) * p->sw + kw;
  p->plane1 = p->WT * p->WH * p->WW;
  p->plane = p->plane1 << ln;
}
