"""ctypes binding of libcoclr_hip.so (see include/coclr_hip.h).

The product path has no fallback: if the shared library is missing or a
kernel call reports an error this module raises.  Only plain pointers, sizes
and the two geometry structs cross the boundary -- no torch types.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("COCLR_LIB_PATH") or os.path.join(_HERE, "libcoclr_hip.so")
ABI_VERSION = 20

i32, i64, f32, f64, vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p


class ConvDesc(C.Structure):
    """Mirror of `coclr_conv_desc`."""
    _fields_ = [(n, i32) for n in (
        "N", "Cin", "Cout", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
        "st", "sh", "sw", "pt", "ph", "pw", "dt", "dh", "dw")] + [
        ("x_nstride", i64), ("y_nstride", i64)] + [(n, i32) for n in (
        "ys_t", "ys_h", "ys_w", "yo_t", "yo_h", "yo_w", "yT", "yH", "yW", "Nx", "algo")]


class PoolDesc(C.Structure):
    """Mirror of `coclr_pool_desc`."""
    _fields_ = [(n, i32) for n in (
        "N", "C", "Ti", "Hi", "Wi", "To", "Ho", "Wo", "kt", "kh", "kw",
        "st", "sh", "sw", "pt", "ph", "pw")] + [("x_nstride", i64), ("y_nstride", i64)]


class ConvCall(C.Structure):
    """Mirror of `coclr_conv_call` (one problem of coclr_conv3d_fwd_multi)."""
    _fields_ = [("d", C.POINTER(ConvDesc)), ("x", vp), ("w_packed", vp), ("y", vp), ("stats", vp),
                ("bias", vp), ("ep_scale", vp), ("ep_shift", vp), ("n_index", vp), ("relu", i32),
                ("accumulate", i32), ("bwd_y", vp), ("bwd_scale", vp), ("bwd_shift", vp), ("bwd_mean", vp),
                ("bwd_invstd", vp), ("bwd_relu", i32), ("in_relu", i32), ("in_scale", vp), ("in_shift", vp)]


class BnFwdCall(C.Structure):
    """Mirror of `coclr_bn_fwd_call`."""
    _fields_ = [(n, vp) for n in ("sum", "sumsq", "gamma", "beta", "running_mean", "running_var",
                                  "num_batches_tracked", "mean", "invstd", "scale", "shift", "y", "z")] + [
        ("count", f64), ("S", i64), ("y_nstride", i64), ("z_nstride", i64), ("C", i32), ("ntiles", i32),
        ("N", i32), ("relu", i32), ("momentum", f32), ("eps", f32)]


class BnBwdCall(C.Structure):
    """Mirror of `coclr_bn_bwd_call`."""
    _fields_ = [(n, vp) for n in ("dz", "y", "scale", "shift", "mean", "invstd", "sums_ws", "dy", "dgamma",
                                  "dbeta")] + [
        ("S", i64), ("dz_nstride", i64), ("y_nstride", i64), ("dy_nstride", i64), ("N", i32), ("C", i32),
        ("relu", i32), ("training", i32), ("part", vp * 2), ("part_ntiles", i32 * 2)]


class GemmEpilogue(C.Structure):
    """Mirror of `coclr_gemm_epilogue`."""
    _fields_ = [("mode", i32), ("S", i32), ("a", vp), ("lda", i64), ("b", vp), ("y", vp), ("inv_norm", vp),
                ("out2", vp), ("f", f32), ("rowsum", vp)]


_P = C.POINTER
_SIGNATURES = {
    "coclr_abi_version": [],
    "coclr_conv_packed_size": [i32, i32, i32, i32, _P(i64)],
    "coclr_conv_pack_weights": [vp, vp, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, i32, i32, vp],
    "coclr_conv_pack_describe": [vp, vp, i32, i32, i32, i64, i64, i32, i32, i32, i32, i32, i32, i32,
                                 _P(i64), _P(i32)],
    "coclr_conv_pack_batch": [vp, vp, i32, vp],
    "coclr_conv3d_ntiles": [_P(ConvDesc), _P(i32)],
    "coclr_conv3d_bwd_sums_ok": [_P(ConvDesc), _P(i32)],
    "coclr_conv3d_fwd": [_P(ConvDesc), vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "coclr_conv3d_fwd_multi": [_P(ConvCall), i32, vp],
    "coclr_conv3d_wgrad_workspace": [_P(ConvDesc), _P(i64)],
    "coclr_conv3d_wgrad": [_P(ConvDesc), vp, vp, vp, vp, i64, i64, i32, i32, vp],
    "coclr_conv3d_wgrad_multi": [_P(ConvDesc), vp, vp, vp, vp, i32, vp, i64, i64, i32, i32, vp],
    "coclr_conv3d_wgrad_bn_ok": [_P(ConvDesc), _P(i32)],
    "coclr_conv3d_wgrad_bn": [_P(ConvDesc), vp, vp, vp, i64, vp, i32, vp, vp, i64, i64, i32, i32, vp],
    "coclr_bn_finalize": [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp],
    "coclr_bn_finalize_apply": [vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp,
                                vp, vp, i32, i64, i64, i64, i32, vp],
    "coclr_bn_finalize_apply_multi": [_P(BnFwdCall), i32, vp],
    "coclr_bn_act_backward_multi": [_P(BnBwdCall), i32, vp],
    "coclr_bn_eval_affine": [vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp],
    "coclr_bn_act_apply": [vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, i64, i32, vp],
    "coclr_bn_backward_workspace": [i32, i32, _P(i64)],
    "coclr_bn_act_backward": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64,
                              i64, i64, i64, i64, i64, i32, i32, i32, vp],
    "coclr_bn_act_backward_coeffs": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i64, i64, i64, i32,
                                     i32, vp],
    "coclr_maxpool3d_fwd": [_P(PoolDesc), vp, vp, vp, vp, vp, i32, vp],
    "coclr_maxpool3d_bwd": [_P(PoolDesc), vp, vp, vp, i64, i64, i32, vp],
    "coclr_bn_act_backward_pooled": [_P(PoolDesc), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64,
                                     i32, i32, vp],
    "coclr_bn_act_backward_pooled_fits": [_P(PoolDesc), _P(i32)],
    "coclr_global_avgpool_fwd": [vp, vp, i64, i64, vp],
    "coclr_global_avgpool_bwd": [vp, vp, i64, i64, vp],
    "coclr_gemm_workspace": [i32, i32, i32, i32, _P(i64)],
    "coclr_gemm": [vp, i64, i64, vp, i64, i64, vp, i64, vp, i32, i32, i32, f32, i32, i32, i32, vp,
                   vp],
    "coclr_gemm_fused": [vp, i64, i64, vp, i64, i64, vp, i64, vp, i32, i32, i32, f32, i32, i32, vp,
                         _P(GemmEpilogue), vp],
    "coclr_l2norm_fwd": [vp, vp, vp, i32, i32, f32, vp],
    "coclr_l2norm_bwd": [vp, vp, vp, vp, i32, i32, vp],
    "coclr_nce_logits_fwd": [vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "coclr_nce_logits_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, i32, vp],
    "coclr_momentum_update": [vp, i32, f32, f32, vp],
    "coclr_queue_enqueue": [vp, vp, i32, i32, i32, vp, vp],
    "coclr_queue_fill_i64": [vp, vp, i64, i32, i32, vp, vp],
    "coclr_queue_advance": [vp, i32, i32, vp],
    "coclr_positive_mask": [vp, vp, vp, vp, i32, i32, i32, vp],
    "coclr_mine_positives": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "coclr_gather_rows": [vp, vp, vp, i32, i64, i64, vp],
    "coclr_pull_rows": [vp, vp, i32, i64, vp],
    "coclr_relu_fwd": [vp, vp, i64, vp],
    "coclr_relu_bwd": [vp, vp, vp, i64, vp],
    "coclr_colsum": [vp, vp, i32, i32, vp],
    "coclr_sigmoid_fwd": [vp, vp, i64, vp],
    "coclr_sigmoid_bwd": [vp, vp, vp, i64, vp],
    "coclr_plane_scale": [vp, vp, vp, vp, i32, i32, i64, i64, i64, i32, vp],
    "coclr_plane_dot": [vp, vp, vp, i32, i32, i64, i64, i64, vp],
    "coclr_adam_step": [vp, i32, vp, vp, vp, i32, f32, f32, vp],
    "coclr_nce_loss_fwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "coclr_nce_loss_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "coclr_stage_clips": [vp, i32, vp, i32, i32, i32, i64, _P(f32), _P(f32), vp],
    "coclr_colstats_workspace": [i32, i32, _P(i64)],
    "coclr_bn1d_stats": [vp, vp, vp, i32, i32, vp],
    "coclr_center_rows": [vp, vp, vp, i32, i32, vp],
    "coclr_retrieval_hits": [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp],
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raise if it cannot be used."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64; import it first so this library binds to the SAME
    # HIP runtime instance (device pointers and streams are shared with torch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            "coclr_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or coclr_amd/csrc/build.sh). There is no fallback path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    got = lib.coclr_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError("coclr_amd: ABI version mismatch (lib %d, python %d); rebuild" %
                              (got, ABI_VERSION))
    _lib = lib
    return lib


CALLS = [0]      # C-ABI calls made so far (bench.py reports calls per step)


def check(rc, what, detail=None):
    """`detail` (e.g. a geometry object) is only formatted when the call failed: these wrappers
    run ~1500 times per training step."""
    CALLS[0] += 1
    if rc != 0:
        if detail is not None:
            what = "%s %s" % (what, detail)
        raise HipLibraryError("coclr_amd: %s failed with hipError %d" % (what, rc))
