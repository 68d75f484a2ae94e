"""Tensor-level wrappers over the C ABI in include/coclr_hip.h.

Every function here takes device tensors (possibly channel-slice views of
wider NCDHW buffers), extracts raw pointers / strides and enqueues the HIP
kernel on torch's *current* stream.  Nothing is computed in Python or by
ATen, and there is no CPU path: host tensors are rejected.
"""
import ctypes as C
import os
import threading

import torch

from . import _lib
from . import plan as _plan
from ._lib import BnBwdCall, BnFwdCall, ConvCall, ConvDesc, PoolDesc


_HANDLE = None
_MAIN_THREAD = threading.get_ident()


def _desc(geom):
    """The launch descriptor of a cached geometry that THIS thread may fill in (strides, sample
    count): geometries are shared and cached, and the forward (caller's thread) and the backward
    (autograd's device thread) of different tensors may use one at the same time -- each side
    writes its own copy."""
    if threading.get_ident() == _MAIN_THREAD:
        return geom.desc
    d = geom.desc_bw
    if d is None:
        d = geom.desc_bw = type(geom.desc).from_buffer_copy(geom.desc)
    return d


def _L():
    """ctypes handle, resolved once (every wrapper below runs ~1500 times per step).  While THIS thread records a
    launch plan (coclr_amd/plan.py) the handle is the recorder's proxy: every entry point runs and is logged."""
    global _HANDLE
    if _HANDLE is None:
        _HANDLE = _lib.load()
    if _plan._ACTIVE is not None:
        rec = _plan.active()
        if rec is not None:
            return rec.proxy
    return _HANDLE


# Diagnostic: COCLR_HOST_DELAY_US=<n> burns n microseconds of host time per kernel launch.  If the step
# time does not move, the host is not on the critical path (tools/host_bound_probe.sh).
_HOST_DELAY = float(os.environ.get("COCLR_HOST_DELAY_US", "0")) * 1e-6


def _stream():
    if _HOST_DELAY:
        import time
        t = time.perf_counter() + _HOST_DELAY
        while time.perf_counter() < t:
            pass
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.HipLibraryError(
            "coclr_amd: kernels need device tensors (got %s); there is no CPU fallback" % t.device)
    if t.dtype != dtype:
        raise TypeError("coclr_amd: expected %s tensor, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def _chk5(t, name):
    """NCDHW view whose (C,T,H,W) part is dense; returns the sample stride."""
    n, c, d, h, w = t.shape
    s = t.stride()
    if not (s[4] == 1 and s[3] == w and s[2] == h * w and s[1] == d * h * w) and t.numel() > 0:
        raise ValueError("coclr_amd: %s must be dense in (C,T,H,W); strides %s shape %s" %
                         (name, s, tuple(t.shape)))
    return s[0] if n > 1 else max(s[0], c * d * h * w)


WINOGRAD = os.environ.get("COCLR_WINOGRAD", "1") != "0"
WINOGRAD_HW = os.environ.get("COCLR_WINOGRAD_HW", "1") != "0"
# (3,1,1) layers through F(4,3) (algo = 2: six contractions per QUAD of output frames, 2x fewer MFMAs than the
# direct form) instead of F(2,3) (algo = 1: four per pair, 1.5x fewer); "0" keeps F(2,3)
WINOGRAD_T4 = os.environ.get("COCLR_WINO_T4", "1") != "0"
WINOGRAD_POLY7 = os.environ.get("COCLR_WINO_POLY7", "1") != "0"
WINOGRAD_PHASES = os.environ.get("COCLR_WINO_PHASES", "1") != "0"
_WINO_HW_MIN = int(os.environ.get("COCLR_WINO_HW_MIN", "16"))      # smallest map side that takes F(2x2,3x3)
_WINO_T4_MIN = int(os.environ.get("COCLR_WINO_T4_MIN", "16"))      # fewest frames that take F(4,3)


def winograd_ok(cin, k, s, p, d, lattice, odim=None, idim=None, algo=1):
    """Can this convolution run through the Winograd kernels (algo = 1)?  Stride-1 'same'
    convolutions of S3D's separable units (backbone/s3dg.py:39-42):
      * (3,1,1), pad (1,0,0): F(2,3) along T -- 4 channel contractions per pair of output frames
        instead of 6;
      * (1,3,3), pad (0,1,1), even H and W >= 4: F(2x2,3x3) -- 16 contractions per 2x2 output block
        instead of 36.
    Narrow layers stay direct (the kernels stage 8-16 channels at a time)."""
    k, p = tuple(k), tuple(p)
    if k == (7, 1, 1):
        # the stride-2 temporal stem conv in polyphase form (F(2,3) on the odd taps + F(2,4) on the even ones:
        # 9 contractions per pair of output frames instead of 14); forward only, even frame counts
        return (WINOGRAD and WINOGRAD_POLY7 and tuple(s) == (2, 1, 1) and p == (3, 0, 0) and
                tuple(d) == (1, 1, 1) and lattice is None and cin >= 16 and odim is not None and
                idim is not None and idim[0] == 2 * odim[0] and odim[0] >= 8)     # >= 4 output pairs per box
    if lattice is not None:
        # a destination lattice along T only, and only in the F(4,3) / F(2,4) kernels (algo = 2): the even / odd
        # phase of the strided temporal stem conv's data gradient (ConvGeom.dgrad_phases)
        ys, yo = tuple(lattice[0]), tuple(lattice[1])
        if not (algo == 2 and ys[1:] == (1, 1) and yo[1:] == (0, 0)):
            return False
    if not (WINOGRAD and tuple(s) == (1, 1, 1) and tuple(d) == (1, 1, 1)):
        return False
    if k == (4, 1, 1):
        return algo == 2 and p == (1, 0, 0) and cin >= 16 and odim is not None and idim is not None and \
            odim[0] == idim[0]
    if k == (3, 1, 1):
        return p == (1, 0, 0) and cin >= 16
    if k == (1, 3, 3):
        return (WINOGRAD_HW and p == (0, 1, 1) and cin >= 16 and odim is not None
                and odim[1] % 2 == 0 and odim[2] % 2 == 0 and odim[1] >= 4 and odim[2] >= 4)
    return False


def winograd_pays(cin, k, s, p, odim, idim=None):
    """Policy of conv_geom(): use Winograd where it is measurably faster.  The F(2x2,3x3) kernel
    holds one workgroup per CU; on 8x8 maps it only ties the direct kernel and on 4x4 maps it
    loses (profiles/r01_g_layers.txt), so the spatial form is kept for maps of 16x16 and up."""
    if not winograd_ok(cin, k, s, p, (1, 1, 1), None, odim, idim):
        return False
    if tuple(k) == (7, 1, 1):
        # measured at the benchmark shape only (32 -> 16 frames: 1.28 -> 1.09 ms, profiles/r06_poly7_ab.txt);
        # shorter clips keep the direct kernel
        return odim[0] >= 16
    return tuple(k) != (1, 3, 3) or (odim[1] >= _WINO_HW_MIN and odim[2] >= _WINO_HW_MIN)


def winograd_t4_pays(odim):
    """F(4,3) instead of F(2,3) for a (3,1,1) layer?  Measured per layer at B=32 (profiles/r06_wino_t4_layers.txt):
    16 frames (Conv_2c.conv2 0.754 -> 0.598 ms forward, Mixed_3c.b1.conv2 0.185 -> 0.145) yes; 8 frames (Mixed_4f
    0.080 -> 0.083) and 4 frames (Mixed_5c 0.050 -> 0.068: one quad per position, a third of the window is padding,
    three workgroups per CU instead of four) no."""
    return WINOGRAD_T4 and odim[0] >= _WINO_T4_MIN


class ConvGeom:
    """Geometry of one 3D convolution (python mirror of coclr_conv_desc)."""

    __slots__ = ("N", "Cin", "Cout", "idim", "odim", "k", "s", "p", "d", "lattice", "algo", "desc",
                 "desc_bw", "_cache")

    def __init__(self, N, Cin, Cout, idim, k, s, p, d=(1, 1, 1), odim=None, lattice=None, algo=0):
        self.N, self.Cin, self.Cout = int(N), int(Cin), int(Cout)
        self.idim = tuple(int(v) for v in idim)
        self.k, self.s, self.p, self.d = (tuple(int(v) for v in t) for t in (k, s, p, d))
        if odim is None:
            odim = tuple(((self.idim[i] - 1) * self.d[i] + 1 + 2 * self.p[i] - self.k[i]) //
                         self.s[i] + 1 for i in range(3))
        self.odim = tuple(int(v) for v in odim)
        if min(self.odim) <= 0:
            raise ValueError("coclr_amd: convolution output would be empty: in %s k %s s %s p %s" %
                             (self.idim, self.k, self.s, self.p))
        # lattice = (step, offset, full dims): output position o lands on element
        # o*step + offset of a tensor of extent `full dims` (one phase of a strided dgrad)
        self.lattice = lattice
        lat = (0,) * 9 if lattice is None else tuple(lattice[0]) + tuple(lattice[1]) + \
            tuple(lattice[2])
        # algo 1 = Winograd (see winograd_ok); the packed operand differs
        self.algo = int(algo)
        if self.algo not in (0, 1, 2):
            raise ValueError("coclr_amd: unknown convolution algorithm %d" % self.algo)
        if self.algo == 2 and self.k not in ((3, 1, 1), (4, 1, 1)):
            raise ValueError("coclr_amd: algorithm 2 is F(4,3) of a (3,1,1) stencil or F(2,4) of a (4,1,1) one")
        if self.algo >= 1 and not winograd_ok(self.Cin, self.k, self.s, self.p, self.d, lattice,
                                              self.odim, self.idim, self.algo):
            raise ValueError("coclr_amd: Winograd needs a (3,1,1) or (1,3,3) stride-1 'same' stencil, or the "
                             "(7,1,1) stride-2 pad-3 temporal stem conv on an even frame count")
        self.desc = ConvDesc(self.N, self.Cin, self.Cout, *self.idim, *self.odim, *self.k,
                             *self.s, *self.p, *self.d, 0, 0, *lat, 0, self.algo)
        self.desc_bw = None
        self._cache = {}

    @property
    def taps(self):
        return self.k[0] * self.k[1] * self.k[2]

    def dgrad(self):
        """Geometry of the data-gradient pass expressed as a forward call:
        stride-1 correlation of the zero-upsampled dY with the flipped stencil."""
        g = self._cache.get("dgrad")
        if g is None:
            pad = tuple(self.k[i] - 1 - self.p[i] for i in range(3))
            if min(pad) < 0:
                raise ValueError("coclr_amd: padding larger than kernel-1 is not supported")
            algo = self.algo if self.algo >= 1 and winograd_ok(self.Cout, self.k, (1, 1, 1), pad,
                                                               self.s, None, self.idim) else 0
            g = ConvGeom(self.N, self.Cout, self.Cin, self.odim, self.k, (1, 1, 1), pad,
                         d=self.s, odim=self.idim, algo=algo)
            self._cache["dgrad"] = g
        return g

    def dgrad_phases(self):
        """Data gradient of a conv strided along ONE axis whose stencil is 1 on the other
        two (S3D's temporal stem conv, backbone/s3dg.py:41), as `stride` dense stride-1
        correlations over dY, each writing one residue class of dX -- no multiply-by-zero
        work, unlike `dgrad()`.  Returns [(geom, tap_base, ntaps, tap_step)] or None when
        the geometry is not of that form."""
        g = self._cache.get("phases", False)
        if g is not False:
            return g
        out = None
        axes = [i for i in range(3) if self.s[i] > 1]
        if len(axes) == 1 and self.d == (1, 1, 1):
            ax = axes[0]
            others = [i for i in range(3) if i != ax]
            if all(self.k[i] == 1 and self.p[i] == 0 for i in others) and ax == 0:
                st, K, P = self.s[ax], self.k[ax], self.p[ax]
                out = []
                for phi in range(st):
                    k0 = (phi + P) % st
                    nk = (K - k0 + st - 1) // st
                    m_phi = (self.idim[ax] - phi + st - 1) // st
                    if nk not in (1, 3, 4) or m_phi <= 0:
                        out = None
                        break
                    c = (phi + P - k0) // st
                    kk, pp, od, ys, yo = [1, 1, 1], [0, 0, 0], list(self.idim), [1, 1, 1], [0, 0, 0]
                    kk[ax], pp[ax], od[ax], ys[ax], yo[ax] = nk, nk - 1 - c, m_phi, st, phi
                    # 3- and 4-tap phases of a long clip through F(4,3) / F(2,4) (algo = 2, lattice-aware kernels)
                    lat = (ys, yo, self.idim)
                    algo = 2 if (WINOGRAD_PHASES and winograd_t4_pays(od) and nk in (3, 4) and
                                 winograd_ok(self.Cout, tuple(kk), (1, 1, 1), tuple(pp), (1, 1, 1), lat,
                                             tuple(od), self.odim, 2)) else 0
                    geom = ConvGeom(self.N, self.Cout, self.Cin, self.odim, kk, (1, 1, 1), pp,
                                    odim=od, lattice=lat, algo=algo)
                    out.append((geom, k0, nk, st))
        self._cache["phases"] = out
        return out

    def ntiles(self):
        v = self._cache.get("ntiles")
        if v is None:
            out = C.c_int32(0)
            _lib.check(_L().coclr_conv3d_ntiles(C.byref(self.desc), C.byref(out)),
                       "conv3d_ntiles", self)
            v = self._cache["ntiles"] = out.value
        return v

    def bwd_sums_ok(self):
        """Can this problem's kernel form BatchNorm backward sums in its epilogue (conv_fwd_multi bwd_bn)?"""
        v = self._cache.get("bwd_sums_ok")
        if v is None:
            out = C.c_int32(0)
            _lib.check(_L().coclr_conv3d_bwd_sums_ok(C.byref(self.desc), C.byref(out)),
                       "conv3d_bwd_sums_ok", self)
            v = self._cache["bwd_sums_ok"] = bool(out.value)
        return v

    def wgrad_bn_ok(self):
        """Is there a weight-gradient kernel that applies the BatchNorm backward of the unit behind this
        convolution while it loads (conv_wgrad_bn)?"""
        v = self._cache.get("wgrad_bn_ok")
        if v is None:
            out = C.c_int32(0)
            _lib.check(_L().coclr_conv3d_wgrad_bn_ok(C.byref(self.desc), C.byref(out)),
                       "conv3d_wgrad_bn_ok", self)
            v = self._cache["wgrad_bn_ok"] = bool(out.value)
        return v

    def wgrad_workspace(self):
        v = self._cache.get("wgws")
        if v is None:
            out = C.c_int64(0)
            _lib.check(_L().coclr_conv3d_wgrad_workspace(C.byref(self.desc), C.byref(out)),
                       "conv3d_wgrad_workspace", self)
            v = self._cache["wgws"] = out.value
        return v

    def __repr__(self):
        return "ConvGeom(N=%d, %d->%d, in=%s, out=%s, k=%s, s=%s, p=%s, d=%s)" % (
            self.N, self.Cin, self.Cout, self.idim, self.odim, self.k, self.s, self.p, self.d)


_GEOMS = {}


def conv_geom(N, Cin, Cout, idim, k, s, p):
    """Shared, cached ConvGeom (its ntiles / dgrad / phase plans are computed once): the engine
    asks for the same ~80 geometries every step.  Picks the algorithm (direct / Winograd)."""
    key = (N, Cin, Cout, tuple(idim), tuple(k), tuple(s), tuple(p))
    g = _GEOMS.get(key)
    if g is None:
        if len(_GEOMS) > 8192:
            _GEOMS.clear()
        g = ConvGeom(N, Cin, Cout, idim, k, s, p)
        if winograd_pays(Cin, k, s, p, g.odim, g.idim):
            g = ConvGeom(N, Cin, Cout, idim, k, s, p,
                         algo=2 if (tuple(k) == (3, 1, 1) and winograd_t4_pays(g.odim)) else 1)
        _GEOMS[key] = g
    return g


def pool_geom(N, Cc, idim, k, s, p):
    key = ("pool", N, Cc, tuple(idim), tuple(k), tuple(s), tuple(p))
    g = _GEOMS.get(key)
    if g is None:
        g = _GEOMS[key] = PoolGeom(N, Cc, idim, k, s, p)
    return g


class PoolGeom:
    __slots__ = ("N", "C", "idim", "odim", "k", "s", "p", "desc", "desc_bw", "_pooled_fits")

    def __init__(self, N, Cc, idim, k, s, p):
        self.N, self.C = int(N), int(Cc)
        self.idim = tuple(int(v) for v in idim)
        self.k, self.s, self.p = (tuple(int(v) for v in t) for t in (k, s, p))
        self.odim = tuple((self.idim[i] + 2 * self.p[i] - self.k[i]) // self.s[i] + 1
                          for i in range(3))
        if min(self.odim) <= 0:
            raise ValueError("coclr_amd: pooling output would be empty")
        self.desc = PoolDesc(self.N, self.C, *self.idim, *self.odim, *self.k, *self.s, *self.p, 0, 0)
        self.desc_bw = None
        self._pooled_fits = None


# ---- convolution ---------------------------------------------------------------

_PACKED = {}


def conv_packed_size(cin, cout, taps, transpose):
    key = (cin, cout, taps, bool(transpose))
    v = _PACKED.get(key)
    if v is not None:
        return v
    v = _PACKED[key] = _conv_packed_size(cin, cout, taps, transpose)
    return v


def _conv_packed_size(cin, cout, taps, transpose):
    out = C.c_int64(0)
    _lib.check(_L().coclr_conv_packed_size(cin, cout, taps, int(transpose), C.byref(out)),
               "conv_packed_size")
    return out.value


def conv_pack_weights(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, transpose,
                      tap_step=1, row0=0, rows_total=0, col0=0, cols_total=0):
    _lib.check(_L().coclr_conv_pack_weights(
        _p(w), _p(packed), cout, cin, taps, co_stride, ci_stride, tap_base, tap_step,
        int(transpose), row0, rows_total, col0, cols_total, _stream()), "conv_pack_weights")


def conv_pack_describe(w, packed, cout, cin, taps, co_stride, ci_stride, tap_base, transpose,
                       tap_step=1, row0=0, rows_total=0, col0=0, cols_total=0):
    """One row of a batch-pack table (16 int64) and the number of 1024-element blocks it takes;
    same arguments as conv_pack_weights."""
    entry = (C.c_int64 * 16)()
    nb = C.c_int32()
    _lib.check(_L().coclr_conv_pack_describe(
        _p(w), _p(packed), cout, cin, taps, co_stride, ci_stride, tap_base, tap_step,
        int(transpose), row0, rows_total, col0, cols_total, entry, C.byref(nb)), "conv_pack_describe")
    return list(entry), nb.value


def conv_pack_batch(table, blockmap, requests=None):
    """Launch every re-layout of a table built from conv_pack_describe rows.  `requests` (the
    argument tuples the rows were made from) is unused here; the CPU test double replays them."""
    _lib.check(_L().coclr_conv_pack_batch(_p(table, torch.int64), _p(blockmap, torch.int32),
                                          blockmap.shape[0], _stream()), "conv_pack_batch")


def conv_fwd(geom, x, w_packed, y, stats=None, bias=None, ep_scale=None, ep_shift=None,
             n_index=None, relu=False, accumulate=False):
    d = _desc(geom)
    d.x_nstride = _chk5(x, "x")
    d.y_nstride = _chk5(y, "y")
    d.Nx = x.shape[0]
    _lib.check(_L().coclr_conv3d_fwd(
        C.byref(d), _p(x), _p(w_packed), _p(y), _p(stats), _p(bias), _p(ep_scale), _p(ep_shift),
        _p(n_index, torch.int64), int(relu), int(accumulate), _stream()), "conv3d_fwd", geom)


def conv_fwd_multi(calls):
    """Independent convolutions in one call (consecutive pairs of the same kernel variant in one launch).
    calls: [dict(geom, x, w, y, stats=None, n_index=None, accumulate=False, bwd_bn=None)]; every problem keeps
    the plan it has alone (statistics: geom.ntiles() slots per channel).  bwd_bn = (y, scale, shift, mean,
    invstd, relu) of a BatchNorm unit: the problem is the data gradient that writes that unit's dz and `stats`
    receives the unit's backward sums (include/coclr_hip.h, coclr_conv_call)."""
    n = len(calls)
    arr = (ConvCall * n)()
    descs = []
    for i, c in enumerate(calls):
        geom, x, y = c["geom"], c["x"], c["y"]
        # every problem gets its OWN descriptor copy: two problems may share one cached geometry
        d = type(geom.desc).from_buffer_copy(_desc(geom))
        d.x_nstride = _chk5(x, "x")
        d.y_nstride = _chk5(y, "y")
        d.Nx = x.shape[0]
        descs.append(d)
        a = arr[i]
        a.d = C.pointer(d)
        a.x, a.w_packed, a.y = _p(x), _p(c["w"]), _p(y)
        a.stats = _p(c.get("stats"))
        a.bias = a.ep_scale = a.ep_shift = None
        a.n_index = _p(c.get("n_index"), torch.int64)
        a.relu = 0
        a.accumulate = int(bool(c.get("accumulate", False)))
        ia = c.get("in_affine")
        if ia is not None:
            # (scale, shift, relu) of the BatchNorm unit whose raw output x is: applied while the kernel reads
            a.in_scale, a.in_shift, a.in_relu = _p(ia[0]), _p(ia[1]), int(bool(ia[2]))
        bb = c.get("bwd_bn")
        if bb is not None:
            # (y, scale, shift, mean, invstd, relu) of the BatchNorm unit whose dz this data gradient writes
            by = bb[0]
            if by.shape != y.shape or _chk5(by, "bwd_y") != d.y_nstride or not y.is_contiguous():
                raise ValueError("coclr_amd: bwd_bn's conv output must be laid out like the destination")
            a.bwd_y, a.bwd_scale, a.bwd_shift = _p(by), _p(bb[1]), _p(bb[2])
            a.bwd_mean, a.bwd_invstd, a.bwd_relu = _p(bb[3]), _p(bb[4]), int(bool(bb[5]))
    _lib.check(_L().coclr_conv3d_fwd_multi(arr, n, _stream()), "conv3d_fwd_multi",
               [c["geom"] for c in calls])


def conv_wgrad(geom, x, dy, dw, workspace, co_stride, ci_stride, tap_base, accumulate=False):
    """dw: the gradient tensor, or a list of up to four tensors that take consecutive blocks of
    output-channel rows (dw[i].shape[0] rows each, summing to geom.Cout)."""
    d = _desc(geom)
    d.x_nstride = _chk5(x, "x")
    d.y_nstride = _chk5(dy, "dy")
    if isinstance(dw, (list, tuple)):
        n = len(dw)
        ptrs = (C.c_void_p * n)(*[_p(t) for t in dw])
        ends, tot = [], 0
        for t in dw:
            tot += t.shape[0]
            ends.append(tot)
        _lib.check(_L().coclr_conv3d_wgrad_multi(
            C.byref(d), _p(x), _p(dy), ptrs, (C.c_int32 * n)(*ends), n, _p(workspace), co_stride,
            ci_stride, tap_base, int(accumulate), _stream()), "conv3d_wgrad_multi", geom)
        return
    _lib.check(_L().coclr_conv3d_wgrad(
        C.byref(d), _p(x), _p(dy), _p(dw), _p(workspace), co_stride, ci_stride, tap_base,
        int(accumulate), _stream()), "conv3d_wgrad", geom)


def conv_wgrad_bn(geom, x, dz, y, coef, relu, dw, workspace, co_stride, ci_stride, tap_base=0,
                  accumulate=False):
    """conv_wgrad over dy = BatchNorm(+ReLU) backward of (dz, y) formed on the fly (coef from
    bn_act_backward_coeffs); geom.wgrad_bn_ok() geometries only."""
    d = _desc(geom)
    d.x_nstride = _chk5(x, "x")
    d.y_nstride = _chk5(dz, "dz")
    _lib.check(_L().coclr_conv3d_wgrad_bn(
        C.byref(d), _p(x), _p(dz), _p(y), _chk5(y, "y"), _p(coef), int(relu), _p(dw), _p(workspace),
        co_stride, ci_stride, tap_base, int(accumulate), _stream()), "conv3d_wgrad_bn", geom)


# ---- batch norm ------------------------------------------------------------------

def bn_finalize(stats, C_, ntiles, count, gamma, beta, running_mean, running_var, nbt, momentum,
                eps, mean, invstd, scale, shift, c0=0, c_total=None):
    """stats: [2][c_total][ntiles] partial sums of a convolution with c_total output channels;
    this call finalises channels [c0, c0+C_)."""
    c_total = C_ if c_total is None else c_total
    base = _p(stats)
    _lib.check(_L().coclr_bn_finalize(
        base + 4 * c0 * ntiles, base + 4 * (c_total + c0) * ntiles, C_, ntiles, float(count),
        _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(nbt, torch.int64), momentum, eps,
        _p(mean), _p(invstd), _p(scale), _p(shift), _stream()), "bn_finalize")


def bn_finalize_apply(stats, C_, ntiles, count, gamma, beta, running_mean, running_var, nbt,
                      momentum, eps, mean, invstd, scale, shift, y, z, relu, c0=0, c_total=None):
    """bn_finalize + bn_act_apply (no residual) in one call; a single launch for small layers.
    y: the [N][C_][S] channel range (a view) the statistics belong to."""
    c_total = C_ if c_total is None else c_total
    base = _p(stats)
    N, _, T, H, W = y.shape
    _lib.check(_L().coclr_bn_finalize_apply(
        base + 4 * c0 * ntiles, base + 4 * (c_total + c0) * ntiles, C_, ntiles, float(count),
        _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(nbt, torch.int64), momentum, eps,
        _p(mean), _p(invstd), _p(scale), _p(shift), _p(y), _p(z), N, T * H * W, _chk5(y, "y"),
        _chk5(z, "z"), int(relu), _stream()), "bn_finalize_apply")


def bn_finalize_apply_multi(units):
    """bn_finalize_apply for several units in one call (one launch for runs of small units).
    units: [dict(stats, C, ntiles, count, bn=(gamma, beta, running_mean, running_var, nbt, momentum, eps),
    small=(mean, invstd, scale, shift), y, z, relu, c0=0, c_total=None)]."""
    n = len(units)
    arr = (BnFwdCall * n)()
    for a, u in zip(arr, units):
        C_, ntiles = u["C"], u["ntiles"]
        c0 = u.get("c0", 0)
        c_total = u.get("c_total") or C_
        base = _p(u["stats"])
        gamma, beta, rm, rv, nbt, momentum, eps = u["bn"]
        mean, invstd, scale, shift = u["small"]
        y, z = u["y"], u["z"]
        N, _, T, H, W = y.shape
        a.sum = base + 4 * c0 * ntiles
        a.sumsq = base + 4 * (c_total + c0) * ntiles
        a.gamma, a.beta, a.running_mean, a.running_var = _p(gamma), _p(beta), _p(rm), _p(rv)
        a.num_batches_tracked = _p(nbt, torch.int64)
        a.mean, a.invstd, a.scale, a.shift = _p(mean), _p(invstd), _p(scale), _p(shift)
        a.y, a.z = _p(y), _p(z)
        a.count = float(u["count"])
        a.S = T * H * W
        a.y_nstride, a.z_nstride = _chk5(y, "y"), _chk5(z, "z")
        a.C, a.ntiles, a.N, a.relu = C_, ntiles, N, int(u["relu"])
        a.momentum, a.eps = momentum, eps
    _lib.check(_L().coclr_bn_finalize_apply_multi(arr, n, _stream()), "bn_finalize_apply_multi")


def bn_act_backward_multi(units):
    """bn_act_backward (no residual) for several units in one call.
    units: [dict(dz, y, scale, shift, mean, invstd, sums, dy, dgamma, dbeta, relu, training, partials=None)];
    partials = [(stats, ntiles)] formed by the data gradient that wrote dz (conv_fwd_multi bwd_bn)."""
    n = len(units)
    arr = (BnBwdCall * n)()
    for a, u in zip(arr, units):
        y = u["y"]
        N, C_, T, H, W = y.shape
        if u["sums"].numel() < 2 * C_ * N:
            raise ValueError("coclr_amd: bn_act_backward workspace too small")
        a.dz, a.y, a.scale, a.shift = _p(u["dz"]), _p(y), _p(u["scale"]), _p(u["shift"])
        a.mean, a.invstd = _p(u["mean"]), _p(u["invstd"])
        a.sums_ws = _p(u["sums"], torch.float64)
        a.dy, a.dgamma, a.dbeta = _p(u["dy"]), _p(u["dgamma"]), _p(u["dbeta"])
        a.S = T * H * W
        a.dz_nstride, a.y_nstride, a.dy_nstride = _chk5(u["dz"], "dz"), _chk5(y, "y"), _chk5(u["dy"], "dy")
        a.N, a.C, a.relu, a.training = N, C_, int(u["relu"]), int(u["training"])
        parts = u.get("partials")
        if parts:
            # [(stats, ntiles)] left by the data gradient(s) that wrote dz: no reduction pass
            if len(parts) > 2:
                raise ValueError("coclr_amd: at most two partial-sum arrays per BatchNorm backward")
            for i, (st, nt) in enumerate(parts):
                if st.numel() != 2 * C_ * nt:
                    raise ValueError("coclr_amd: partial sums do not match the unit's channels")
                a.part[i] = _p(st)
                a.part_ntiles[i] = nt
    _lib.check(_L().coclr_bn_act_backward_multi(arr, n, _stream()), "bn_act_backward_multi")


SMALL_CHANNEL = 32768      # N*S per channel up to which the one-launch BatchNorm forms are used


def bn_eval_affine(gamma, beta, running_mean, running_var, eps, C_, mean, invstd, scale, shift):
    _lib.check(_L().coclr_bn_eval_affine(
        _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, C_, _p(mean), _p(invstd),
        _p(scale), _p(shift), _stream()), "bn_eval_affine")


def bn_act_apply(y, scale, shift, residual, z, relu):
    N, C_, T, H, W = y.shape
    S = T * H * W
    _lib.check(_L().coclr_bn_act_apply(
        _p(y), _p(scale), _p(shift), _p(residual), _p(z), N, C_, S, _chk5(y, "y"), _chk5(z, "z"),
        _chk5(residual, "residual") if residual is not None else 0, int(relu), _stream()),
        "bn_act_apply")


def bn_backward_workspace(N, C_):
    """float64 elements of the partial-sum workspace of bn_act_backward."""
    return 2 * C_ * N


def bn_act_backward(dz, y, z, scale, shift, mean, invstd, sums_ws, dy, dres, dgamma, dbeta, relu,
                    training, dres_accumulate=False):
    N, C_, T, H, W = y.shape
    S = T * H * W
    if sums_ws.numel() < 2 * C_ * N:
        raise ValueError("coclr_amd: bn_act_backward workspace too small")
    _lib.check(_L().coclr_bn_act_backward(
        _p(dz), _p(y), _p(z), _p(scale), _p(shift), _p(mean), _p(invstd),
        _p(sums_ws, torch.float64), _p(dy), _p(dres), _p(dgamma), _p(dbeta), N, C_, S,
        _chk5(dz, "dz"), _chk5(y, "y"), _chk5(dy, "dy"), _chk5(z, "z") if z is not None else 0,
        _chk5(dres, "dres") if dres is not None else 0, int(relu), int(training),
        int(dres_accumulate), _stream()), "bn_act_backward")


def bn_act_backward_coeffs(dz, y, scale, shift, mean, invstd, sums_ws, coef, dgamma, dbeta, relu, training):
    """The reduction pass of bn_act_backward and coef[5][C] (A, B, D, scale, shift) of its apply pass, which
    the reader of dy runs itself (conv_wgrad_bn)."""
    N, C_, T, H, W = y.shape
    if sums_ws.numel() < 2 * C_ * N:
        raise ValueError("coclr_amd: bn_act_backward workspace too small")
    if coef.numel() < 5 * C_:
        raise ValueError("coclr_amd: bn_act_backward_coeffs needs 5 x C coefficients")
    _lib.check(_L().coclr_bn_act_backward_coeffs(
        _p(dz), _p(y), _p(scale), _p(shift), _p(mean), _p(invstd), _p(sums_ws, torch.float64), _p(coef),
        _p(dgamma), _p(dbeta), N, C_, T * H * W, _chk5(dz, "dz"), _chk5(y, "y"), int(relu), int(training),
        _stream()), "bn_act_backward_coeffs")


# ---- pooling ---------------------------------------------------------------------

def maxpool_fwd(geom, x, y, indices=None, in_scale=None, in_shift=None, in_relu=False):
    """in_scale / in_shift: per-channel affine (+ ReLU) applied to x as it is read -- the pool then
    consumes the RAW convolution output of the BatchNorm unit in front of it."""
    d = _desc(geom)
    d.x_nstride = _chk5(x, "x")
    d.y_nstride = _chk5(y, "y")
    _lib.check(_L().coclr_maxpool3d_fwd(C.byref(d), _p(x), _p(y), _p(indices, torch.int32),
                                        _p(in_scale), _p(in_shift), int(in_relu), _stream()),
               "maxpool3d_fwd")


def maxpool_bwd(geom, dy, indices, dx, accumulate=False):
    _lib.check(_L().coclr_maxpool3d_bwd(
        C.byref(_desc(geom)), _p(dy), _p(indices, torch.int32), _p(dx), _chk5(dy, "dy"),
        _chk5(dx, "dx"), int(accumulate), _stream()), "maxpool3d_bwd")


def bn_act_backward_pooled(geom, pool_dy, indices, y, scale, shift, mean, invstd, sums_ws, dy, dgamma,
                           dbeta, relu, training):
    """BatchNorm(+ReLU) backward of a unit consumed in fused form by the max-pool `geom`
    (maxpool_fwd with in_scale / in_shift): pool backward, the two BatchNorm reductions and the
    apply pass without ever writing the gradient of the normalised activation."""
    N, C_ = y.shape[0], y.shape[1]
    if sums_ws.numel() < 2 * C_ * N:
        raise ValueError("coclr_amd: bn_act_backward_pooled workspace too small")
    _lib.check(_L().coclr_bn_act_backward_pooled(
        C.byref(_desc(geom)), _p(pool_dy), _p(indices, torch.int32), _p(y), _p(scale), _p(shift),
        _p(mean), _p(invstd), _p(sums_ws, torch.float64), _p(dy), _p(dgamma), _p(dbeta),
        _chk5(pool_dy, "pool_dy"), _chk5(y, "y"), _chk5(dy, "dy"), int(relu), int(training),
        _stream()), "bn_act_backward_pooled")


def pooled_backward_fits(geom):
    """Does bn_act_backward_pooled take this pool?  Asked of the library (the answer depends on the
    kernel's LDS tiling), once per geometry."""
    v = getattr(geom, "_pooled_fits", None)
    if v is None:
        out = C.c_int32(0)
        _lib.check(_L().coclr_bn_act_backward_pooled_fits(C.byref(geom.desc), C.byref(out)),
                   "bn_act_backward_pooled_fits")
        v = bool(out.value)
        try:
            geom._pooled_fits = v
        except AttributeError:
            pass
    return v


def global_avgpool_fwd(x, y):
    planes = x.shape[0] * x.shape[1]
    _lib.check(_L().coclr_global_avgpool_fwd(_p(x), _p(y), planes, x.numel() // planes,
                                                    _stream()), "global_avgpool_fwd")


def global_avgpool_bwd(dy, dx):
    planes = dx.shape[0] * dx.shape[1]
    _lib.check(_L().coclr_global_avgpool_bwd(_p(dy), _p(dx), planes, dx.numel() // planes,
                                                    _stream()), "global_avgpool_bwd")


# ---- head --------------------------------------------------------------------------

def gemm_workspace(M, N, K, splits):
    out = C.c_int64(0)
    _lib.check(_L().coclr_gemm_workspace(M, N, K, splits, C.byref(out)), "gemm_workspace")
    return out.value


def gemm(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, alpha=1.0, relu=False, accumulate=False,
         splits=1, workspace=None):
    _lib.check(_L().coclr_gemm(
        _p(a), sam, sak, _p(b), sbk, sbn, _p(c), ldc, _p(bias), M, N, K, alpha, int(relu),
        int(accumulate), splits, _p(workspace), _stream()), "gemm")


def gemm_fused_workspace(M, N, K, splits):
    return M * N * max(1, splits)


def gemm_fused(a, sam, sak, b, sbk, sbn, c, ldc, bias, M, N, K, alpha=1.0, relu=False, splits=1,
               workspace=None, mode=0, S=0, ep_a=None, lda=0, ep_b=None, ep_y=None, inv_norm=None, out2=None,
               f=0.0, rowsum=None):
    """coclr_gemm_fused (include/coclr_hip.h): the product with the row-level operation that follows it
    (mode 1 ReLU backward, 2 row normalise, 3 l_pos + normalise backward, 4 average-pool backward) applied
    by the kernel that folds the split-K partials; mode 0 with splits 1 is the plain product in one launch
    (optionally + row sums of A)."""
    ep = _lib.GemmEpilogue(int(mode), int(S), _p(ep_a), int(lda), _p(ep_b), _p(ep_y), _p(inv_norm), _p(out2),
                           float(f), _p(rowsum))
    _lib.check(_L().coclr_gemm_fused(
        _p(a), sam, sak, _p(b), sbk, sbn, _p(c), ldc, _p(bias), M, N, K, alpha, int(relu), splits,
        _p(workspace), C.byref(ep), _stream()), "gemm_fused")


def l2norm_fwd(x, y, inv_norm, eps=1e-12):
    rows, D = x.shape
    _lib.check(_L().coclr_l2norm_fwd(_p(x), _p(y), _p(inv_norm), rows, D, eps, _stream()),
               "l2norm_fwd")


def l2norm_bwd(dy, y, inv_norm, dx):
    rows, D = y.shape
    _lib.check(_L().coclr_l2norm_bwd(_p(dy), _p(y), _p(inv_norm), _p(dx), rows, D,
                                            _stream()), "l2norm_bwd")


def nce_logits_fwd(q, k, queue, logits, T):
    B, D = q.shape
    K = queue.shape[1]
    _lib.check(_L().coclr_nce_logits_fwd(_p(q), _p(k), _p(queue), _p(logits), B, D, K, T,
                                                _stream()), "nce_logits_fwd")


def nce_logits_bwd(dlogits, k, queue, dq, workspace, T, splits):
    B, D = dq.shape
    K = queue.shape[1]
    _lib.check(_L().coclr_nce_logits_bwd(_p(dlogits), _p(k), _p(queue), _p(dq),
                                                _p(workspace), B, D, K, T, splits, _stream()),
               "nce_logits_bwd")


def momentum_update(table, nchunks, m, one_minus_m, pairs=None):
    """`pairs` = (dst tensors, src tensors) the pointer table was built from: not used by the
    kernel; callers pass it so the tensors stay referenced while the launch is queued."""
    _lib.check(_L().coclr_momentum_update(_p(table, torch.int64), nchunks, m, one_minus_m,
                                                 _stream()), "momentum_update")


def queue_enqueue(queue, keys, ptr):
    D, K = queue.shape
    BW = keys.shape[0]
    _lib.check(_L().coclr_queue_enqueue(_p(queue), _p(keys), D, K, BW,
                                               _p(ptr, torch.int64), _stream()), "queue_enqueue")


def queue_fill_i64(queue, vals, const_val, BW, ptr):
    _lib.check(_L().coclr_queue_fill_i64(
        _p(queue, torch.int64), _p(vals, torch.int64), const_val, queue.shape[0], BW,
        _p(ptr, torch.int64), _stream()), "queue_fill_i64")


def queue_advance(ptr, BW, K):
    _lib.check(_L().coclr_queue_advance(_p(ptr, torch.int64), BW, K, _stream()),
               "queue_advance")


def positive_mask(sim, src, names, mask, topk):
    B, K1 = mask.shape
    _lib.check(_L().coclr_positive_mask(
        _p(sim), _p(src, torch.int64), _p(names, torch.int64), _p(mask, torch.uint8), B, K1 - 1,
        topk, _stream()), "positive_mask")


def mine_workspace(B, K, topk, device):
    """(candidate values, candidate columns, zeroed row-tile counters) for mine_positives; the
    counters are left zero by every launch, so one workspace serves a model for its lifetime."""
    nt = (K + 63) // 64
    return (torch.empty(B * nt * max(topk, 1), dtype=torch.float32, device=device),
            torch.empty(B * nt * max(topk, 1), dtype=torch.int32, device=device),
            torch.zeros((B + 31) // 32, dtype=torch.int32, device=device))


def mine_positives(kf, queue_second, src, names, mask, topk, workspace, sim_out=None):
    """mask <- column 0 | same-source columns | top-k of kf @ queue_second over the other columns, in
    one launch (model/pretrain.py:397-413)."""
    B, D = kf.shape
    K = queue_second.shape[1]
    cv, ci, cnt = workspace
    nt = (K + 63) // 64
    if cv.numel() < B * nt * max(topk, 1) or cnt.numel() < (B + 31) // 32:
        raise ValueError("coclr_amd: mine_positives workspace too small")
    _lib.check(_L().coclr_mine_positives(
        _p(kf), _p(queue_second), _p(src, torch.int64), _p(names, torch.int64), _p(mask, torch.uint8),
        _p(cv), _p(ci, torch.int32), _p(cnt, torch.int32), _p(sim_out), B, D, K, topk, _stream()),
        "mine_positives")


def gather_rows(inp, idx, out):
    """out[i] = inp[idx[i]]; `inp` rows must be dense but may be strided along dim 0."""
    rows = out.shape[0]
    row_elems = out.numel() // rows
    if inp.dim() > 1 and inp[0].numel() == row_elems and inp[0].is_contiguous():
        in_stride = inp.stride(0) if inp.shape[0] > 1 else row_elems
    else:
        raise ValueError("coclr_amd: gather_rows needs dense source rows")
    _lib.check(_L().coclr_gather_rows(_p(inp), _p(idx, torch.int64), _p(out), rows,
                                             row_elems, max(in_stride, row_elems), _stream()),
               "gather_rows")


def pull_rows(row_ptrs, out, keep=None):
    """out[i] = row at device address row_ptrs[i] (int64 tensor); `keep`: the (peer) tensors the
    addresses point into, referenced while the launch is queued."""
    rows = out.shape[0]
    _lib.check(_L().coclr_pull_rows(_p(row_ptrs, torch.int64), _p(out), rows, out.numel() // rows,
                                    _stream()), "pull_rows")


def relu_fwd(x, y):
    _lib.check(_L().coclr_relu_fwd(_p(x), _p(y), x.numel(), _stream()), "relu_fwd")


def relu_bwd(dy, y, dx):
    _lib.check(_L().coclr_relu_bwd(_p(dy), _p(y), _p(dx), y.numel(), _stream()), "relu_bwd")


def colsum(x, out):
    rows, cols = x.shape
    _lib.check(_L().coclr_colsum(_p(x), _p(out), rows, cols, _stream()), "colsum")


# ---- S3D-G self gating ---------------------------------------------------------------

def sigmoid_fwd(s_, w):
    _lib.check(_L().coclr_sigmoid_fwd(_p(s_), _p(w), s_.numel(), _stream()), "sigmoid_fwd")


def sigmoid_bwd(dw, w, ds):
    _lib.check(_L().coclr_sigmoid_bwd(_p(dw), _p(w), _p(ds), w.numel(), _stream()), "sigmoid_bwd")


def plane_scale(a, gain, bias, out, accumulate=False):
    """out[n][c][:] (+)= a[n][c][:] * gain[n][c] + bias[n][c]; a / out may be channel slices."""
    N, C_, T, H, W = a.shape
    _lib.check(_L().coclr_plane_scale(_p(a), _p(gain), _p(bias), _p(out), N, C_, T * H * W,
                                      _chk5(a, "a"), _chk5(out, "out"), int(accumulate), _stream()),
               "plane_scale")


def plane_dot(a, b, out):
    """out[n][c] = <a[n][c], b[n][c]> over (T, H, W)."""
    N, C_, T, H, W = a.shape
    _lib.check(_L().coclr_plane_dot(_p(a), _p(b), _p(out), N, C_, T * H * W, _chk5(a, "a"),
                                    _chk5(b, "b"), _stream()), "plane_dot")


# ---- training-loop neighbours: optimiser, loss epilogue, input staging -------------------

def adam_step(table, nchunks, hyper, steps, groups, ngroups, mom_m=0.0, mom_1m=0.0, keep=None):
    """One multi-tensor Adam launch (+ folded momentum-encoder update where the table names a key
    parameter).  `keep`: the tensors the table points at, referenced while the launch is queued."""
    _lib.check(_L().coclr_adam_step(_p(table, torch.int64), nchunks, _p(hyper, torch.float64), _p(steps),
                                    _p(groups, torch.int32), ngroups, mom_m, mom_1m, _stream()),
               "adam_step")


def nce_loss_fwd(logits, mask, target, rowstats, flags, scalars, mode, drop_self=False, k1=1, k2=5):
    B, N1 = logits.shape
    _lib.check(_L().coclr_nce_loss_fwd(
        _p(logits), _p(mask, torch.uint8), _p(target, torch.int64), _p(rowstats),
        _p(flags, torch.uint8), _p(scalars), B, N1, mode, int(drop_self), k1, k2, _stream()),
        "nce_loss_fwd")


def nce_loss_bwd(logits, mask, target, rowstats, flags, dloss, dlogits, mode):
    B, N1 = logits.shape
    _lib.check(_L().coclr_nce_loss_bwd(
        _p(logits), _p(mask, torch.uint8), _p(target, torch.int64), _p(rowstats),
        _p(flags, torch.uint8), _p(dloss), _p(dlogits), B, N1, mode, _stream()), "nce_loss_bwd")


def stage_clips(frames, out, S, mean, std):
    """frames (B, C, S*T, H, W) uint8 or fp32 in [0,1] -> out (B, S, C, T, H, W) fp32."""
    B, Cc = frames.shape[0], frames.shape[1]
    thw = frames[0, 0].numel() // S
    if frames.dtype == torch.uint8:
        src, u8 = _p(frames, torch.uint8), 1
    else:
        src, u8 = _p(frames), 0
    m = (C.c_float * Cc)(*[float(v) for v in mean])
    s = (C.c_float * Cc)(*[float(v) for v in std])
    _lib.check(_L().coclr_stage_clips(src, u8, _p(out), B, Cc, S, thw, m, s, _stream()),
               "stage_clips")


# ---- evaluation consumers ---------------------------------------------------------------

def colstats_workspace(rows, cols):
    out = C.c_int64(0)
    _lib.check(_L().coclr_colstats_workspace(rows, cols, C.byref(out)), "colstats_workspace")
    return out.value


def bn1d_stats(x, stats, workspace):
    rows, cols = x.shape
    _lib.check(_L().coclr_bn1d_stats(_p(x), _p(stats), _p(workspace), rows, cols, _stream()),
               "bn1d_stats")


def center_rows(x, out, workspace):
    rows, cols = x.shape
    _lib.check(_L().coclr_center_rows(_p(x), _p(out), _p(workspace), rows, cols, _stream()),
               "center_rows")


def retrieval_hits(sim, train_label, test_label, ks, hits, topidx=None):
    """ks: int32 device tensor, ascending; hits fp32 (B, len(ks)); topidx int32 (B, ks[-1])."""
    B, N = sim.shape
    if topidx is None:
        raise ValueError("coclr_amd: retrieval_hits needs the topidx buffer (B, kmax)")
    kmax = int(topidx.shape[1])
    _lib.check(_L().coclr_retrieval_hits(
        _p(sim), _p(train_label, torch.int64), _p(test_label, torch.int64), _p(ks, torch.int32),
        ks.shape[0], _p(hits), _p(topidx, torch.int32), B, N, kmax, _stream()),
        "retrieval_hits")
