"""Launch plans: the C-ABI calls of one module pass, recorded once and re-issued with pre-marshalled arguments.

A differentiated pass of a backbone node is ~240 calls of libcoclr_hip.so forward and as many backward, each
behind ~30 us of Python (generators, geometry look-ups, torch.empty, ctypes marshalling): 11-16 ms of host
work per step.  The device does not wait for it while the host runs ahead -- but the reference's own loop
synchronises three times per iteration (`top1.item()`, `top5.item()`, `loss.item()`, main_nce.py:320-327), and
after every synchronisation the launch-dense 8x8x8 / 4x4x4 stages of the backward pass start with no lead.

The launch sequence of a pass is static per (module, input shape, parameter / buffer / bucket addresses), so:
the pass is run ONCE with its allocations routed to a private memory pool (torch.cuda.MemPool: every activation,
gradient and workspace then has an address nobody else is given) while `Recorder` logs every C-ABI call it
makes -- function pointer plus arguments frozen as ctypes objects, descriptors copied -- and every stream
dependency (engine._note).  Every later pass re-issues the log: same kernels, same operands, same streams, same
order -- eager launches, not a hipGraph (whose replay serialises the weight-gradient branch on ROCm: -5 % on
the device, docs/history.md section 2) -- at ~2 us of host time per call.  Results are bit-identical to the
interpreted pass (tests/test_gpu_model.py::test_planned_query_encoder_matches_eager).
"""
import ctypes as C
import threading

from . import _lib

_ACTIVE = None          # the Recorder of the thread that is recording, if any
# Measurement hook (bench.py): PROBE(key, start event, end event) receives HIP events recorded around the entries a
# caller marked while the plan was recorded (LaunchPlan.marks) -- the re-issued launches of one kernel cannot be
# bracketed from outside any more.  None (the default): replay does not look at the marks.
PROBE = None

# host-side queries (no launch, results cached by their callers): never part of a plan
_QUERIES = frozenset((
    "coclr_abi_version", "coclr_conv_packed_size", "coclr_conv_pack_describe", "coclr_conv3d_ntiles",
    "coclr_conv3d_bwd_sums_ok", "coclr_conv3d_wgrad_bn_ok", "coclr_conv3d_wgrad_workspace", "coclr_bn_backward_workspace",
    "coclr_bn_act_backward_pooled_fits", "coclr_gemm_workspace", "coclr_colstats_workspace"))

_CARG = type(C.byref(C.c_int()))


def active():
    """The recorder of THIS thread (the forward is recorded on the caller's thread, the backward on autograd's)."""
    r = _ACTIVE
    if r is not None and r.tid == threading.get_ident():
        return r
    return None


class LaunchPlan:
    """An ordered log of (C function, frozen arguments) and of Python callables (stream dependencies)."""

    __slots__ = ("entries", "keep", "ncalls", "stream", "marks")

    def __init__(self, stream):
        self.entries = []      # (fn, args list) | (None, callable)
        self.keep = []         # ctypes objects the frozen arguments point into
        self.ncalls = 0
        self.stream = stream   # handle of the stream the pass was recorded on (replay must be on it)
        self.marks = []        # (first entry, one past the last entry, key): ranges a measurement brackets

    def replay(self):
        probe = PROBE
        if probe is not None and self.marks:
            return self._replay_probed(probe)
        for fn, args in self.entries:
            if fn is None:
                args()
            else:
                rc = fn(*args)
                if rc != 0:
                    raise _lib.HipLibraryError("coclr_amd: %s failed with hipError %d (launch plan replay)"
                                               % (getattr(fn, "__name__", "a C-ABI call"), rc))
        _lib.CALLS[0] += self.ncalls

    def _replay_probed(self, probe):
        """replay() with HIP events on the current stream around the marked ranges."""
        import torch
        begins = {b: k for b, e, k in self.marks}
        ends = {e: k for b, e, k in self.marks}
        pending = {}
        for i, (fn, args) in enumerate(self.entries):
            if i in begins:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                pending[begins[i]] = ev
            if fn is None:
                args()
            else:
                rc = fn(*args)
                if rc != 0:
                    raise _lib.HipLibraryError("coclr_amd: %s failed with hipError %d (launch plan replay)"
                                               % (getattr(fn, "__name__", "a C-ABI call"), rc))
            if i + 1 in ends and ends[i + 1] in pending:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                probe(ends[i + 1], pending.pop(ends[i + 1]), ev)
        _lib.CALLS[0] += self.ncalls

    def pointer_refs(self, lo, hi):
        """[(entry index, argument index, offset)] of the top-level pointer arguments that point into
        [lo, hi): the places to patch when the tensor that lived there at recording time moves."""
        refs = []
        for i, (fn, args) in enumerate(self.entries):
            if fn is None:
                continue
            for j, a in enumerate(args):
                if isinstance(a, C.c_void_p) and a.value is not None and lo <= a.value < hi:
                    refs.append((i, j, a.value - lo))
        return refs

    def embedded_refs(self, lo, hi):
        """Does any struct handed over by pointer (multi-problem calls) hold an address in [lo, hi)?  Those are
        not patched; a caller that finds one falls back to a static input copy."""
        for obj in self.keep:
            try:
                raw = bytes(obj)
            except TypeError:
                continue
            for off in range(0, len(raw) - 7, 8):
                v = int.from_bytes(raw[off:off + 8], "little")
                if lo <= v < hi:
                    return True
        return False

    def patch(self, refs, base):
        for i, j, off in refs:
            self.entries[i][1][j] = C.c_void_p(base + off)


class _Proxy:
    """Stands in for the ctypes library handle while a pass is recorded: every entry point runs AND is logged."""

    def __init__(self, rec, lib):
        self._rec, self._lib = rec, lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in _QUERIES:
            setattr(self, name, fn)
            return fn
        argtypes = fn.argtypes
        rec = self._rec

        def call(*args):
            frozen = [_freeze(t, a, rec.plan.keep) for t, a in zip(argtypes, args)]
            if len(frozen) != len(argtypes):
                raise TypeError("coclr_amd: %s takes %d arguments, got %d" % (name, len(argtypes), len(args)))
            rec.plan.entries.append((fn, frozen))
            rec.plan.ncalls += 1
            return fn(*frozen)

        setattr(self, name, call)
        return call


def _freeze(argtype, a, keep):
    """`a` as an object ctypes passes without looking at Python state again; what it points to is copied when
    somebody else may write to it later (the shared, cached geometry descriptors)."""
    if a is None:
        return None
    if isinstance(a, _CARG):                    # byref(struct): the struct may be a shared descriptor
        obj = a._obj
        cp = type(obj).from_buffer_copy(obj)
        keep.append(obj)                        # what ITS pointers point to stays alive
        keep.append(cp)
        return C.byref(cp)
    if isinstance(a, (C.Array, C.Structure, C._Pointer)):
        keep.append(a)                          # per-call tables: built for this call, not written again
        return a
    if isinstance(a, C._SimpleCData):
        return a
    return argtype(a)


class Recorder:
    def __init__(self, stream):
        self.tid = threading.get_ident()
        self.plan = LaunchPlan(stream)
        self.proxy = _Proxy(self, _lib.load())
        self.tainted = None     # why the pass cannot be replayed (a launch the log does not see)

    def py(self, fn):
        self.plan.entries.append((None, fn))

    def taint(self, why):
        if self.tainted is None:
            self.tainted = why

    def __enter__(self):
        global _ACTIVE
        if _ACTIVE is not None:
            raise RuntimeError("coclr_amd: a launch plan is already being recorded")
        _ACTIVE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE
        _ACTIVE = None
        return False
