"""DistributedDataParallel default for this hot path.

The launch scripts wrap the model as `DistributedDataParallel(model, device_ids=[gpu])`
(main_nce.py:172, main_coclr.py:184).  With torch's default `gradient_as_bucket_view=False` DDP copies
every gradient INTO its bucket when it becomes ready and BACK out after the all-reduce: 235 + 235
tiny launches per step on the critical stream, ~1 ms of a 37 ms step on MI355X
(profiles/r02_serial_kernel_stats.csv: `__amd_rocclr_copyBuffer`).  With
`gradient_as_bucket_view=True` the gradients ARE views of the buckets and the copy-back disappears.
The flag only constrains callers that `detach_()` gradients; the launch scripts never touch
`.grad` (they call `zero_grad()`, `backward()`, `step()`), so for modules of THIS package the shim
makes True the default when the caller did not say otherwise.  `COCLR_PATCH_DDP=0` opts out.

With the gradients living in the buckets, the remaining per-parameter work is DDP's copy INTO the
bucket (`mul_out(bucket_view, grad, 1/world)`: 235 launches of ~3 us on the critical stream, even at
world size 1).  The shim therefore also registers a communication hook on wrappers it defaulted
(`COCLR_DDP_HOOK=0` opts out; nothing is registered when the caller chose the flag) -- at the wrapper's
FIRST forward, and only if the caller has not registered a hook of their own by then (DDP accepts exactly
one); the slots it publishes to the engine are dropped when the wrapper is garbage-collected: the hook
  * tells the engine where each parameter's gradient lives in the bucket (`engine.set_grad_slot`), so
    the next backward writes weight gradients straight into the bucket views and DDP, finding them
    there, copies nothing;
  * averages per BUCKET: one `div_` over the flat buffer + one all-reduce (RCCL) -- bit-identical to
    DDP's per-parameter pre-division for power-of-two world sizes, nothing at all at world size 1.
"""
import inspect
import os
import sys

import torch

_DDP = torch.nn.parallel.DistributedDataParallel


def _positional_index():
    """Index of gradient_as_bucket_view among the positional arguments after `module` (9 on torch
    >= 2.4, which has init_sync in front of it; 8 before)."""
    names = [n for n in inspect.signature(_DDP.__init__).parameters if n not in ("self", "module")]
    if "gradient_as_bucket_view" not in names:      # somebody wrapped __init__ with (*args, **kwargs)
        return 9
    return names.index("gradient_as_bucket_view")


_POS = _positional_index()
_installed = [False]

# ---- collectives: one choke point ---------------------------------------------------------------------
# Every cross-rank call of the hot path goes through `collective(name, fn)`:
#   * LAST holds (name, monotonic time, sequence number) of the most recent call: a watchdog (bench.py)
#     that sees no progress can say WHICH exchange a rank is stuck in instead of timing out silently;
#   * an exception out of the backend (gloo / RCCL timeout, peer gone) is re-raised naming the call site;
#   * with TIMINGS set to a list, each call is bracketed by device synchronisations and its wall time is
#     recorded as (name, bytes, milliseconds): an instrumented, serialised step -- never the timed region.
LAST = ["", 0.0, 0]
TIMINGS = None
_PREP = {}            # device -> stream on which bucket pre-division + all-reduce are enqueued


def collective(name, fn, nbytes=0, device=None):
    import time
    LAST[0], LAST[1], LAST[2] = name, time.monotonic(), LAST[2] + 1
    timing = TIMINGS
    if timing is not None and device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    try:
        out = fn()
        if timing is not None:
            if hasattr(out, "wait"):
                out.wait()
            if device is not None and device.type == "cuda":
                torch.cuda.synchronize(device)
            timing.append((name, int(nbytes), (time.perf_counter() - t0) * 1e3))
    except Exception as e:
        msg = "coclr_amd: collective '%s' failed after %.1f s: %s" % (name, time.perf_counter() - t0, e)
        # keep the backend's exception TYPE (DistBackendError / DistNetworkError are what elastic launchers
        # catch to restart); anything that cannot be rebuilt from a message becomes a RuntimeError
        err = None
        if isinstance(e, RuntimeError):
            try:
                err = type(e)(msg)
            except Exception:
                err = None
        raise (err if err is not None else RuntimeError(msg)) from e
    return out



class _HookState:
    def __init__(self, group):
        self.group = group
        self.seen = {}        # bucket index -> (buffer address, #parameters) already published
        self.published = {}   # id(param) -> (weakref(param), address of the view handed to the engine)
        self.verified = set() # bucket indices whose published views DDP has been seen to accept


def _drop_slots(published):
    """The wrapper is gone (re-wrap, or training continues on the bare model): its bucket storage must not
    stay alive -- or keep being handed out as `.grad` memory -- through the engine's slot table.  Only slots
    that still hold THIS wrapper's view are dropped: the cyclic gc may finalise a dead wrapper long after
    a new wrapper around the same model has published its own views for the same parameters."""
    from . import engine
    for ref, addr in published.values():
        p = ref()
        if p is None:
            continue
        slot = engine._GRAD_SLOTS.get(id(p))
        if slot is not None and slot[0]() is p and slot[1].data_ptr() == addr:
            engine.set_grad_slot(p, None)
    published.clear()


def _still_published(state, params, grads):
    """Does the engine still hold the views this wrapper published for the bucket?  (Checked on the first and
    the last parameter: slots are dropped or replaced a whole wrapper at a time.)"""
    from . import engine
    for i in (0, len(params) - 1):
        slot = engine._GRAD_SLOTS.get(id(params[i]))
        if slot is None or slot[0]() is not params[i] or slot[1].data_ptr() != grads[i].data_ptr():
            return False
    return True


def _register_lazily(ddp, _inputs):
    """Forward pre-hook of a wrapper this shim defaulted: at its FIRST forward, register the bucket hook
    unless the caller has registered a communication hook of their own meanwhile (DDP takes exactly
    one: fp16 compression, PowerSGD, ... win, and the engine then simply gets no bucket views)."""
    import weakref
    st = ddp.__dict__.pop("_coclr_hook_pending", None)
    if st is None:
        return
    state, handle = st
    handle.remove()
    if getattr(ddp, "_comm_hooks", None):
        return
    try:
        ddp.register_comm_hook(state, bucket_hook)
    except RuntimeError:
        return                              # a hook is already there
    weakref.finalize(ddp, _drop_slots, state.published)


def bucket_hook(state, bucket):
    """DDP communication hook: publish the bucket's gradient views to the engine, then average the
    flat buffer (what DDP's built-in path computes, per bucket instead of per parameter)."""
    import torch.distributed as dist
    from . import engine
    buf = bucket.buffer()
    sig = (buf.data_ptr(), buf.numel())
    idx = bucket.index()
    params, grads = bucket.parameters(), bucket.gradients()
    if state.seen.get(idx) != sig or not _still_published(state, params, grads):
        # new bucket memory -- or somebody (another wrapper around the same model, a late finaliser of a dead
        # one) has taken the engine's slots since: publish again, verification starts over
        import weakref
        for p, g in zip(params, grads):
            engine.set_grad_slot(p, g)
            state.published[id(p)] = (weakref.ref(p), g.data_ptr())
        state.seen[idx] = sig
        state.verified.discard(idx)
    elif idx not in state.verified:
        # same bucket memory as last time: if every gradient DDP holds for it IS the view the engine was
        # given, DDP copied nothing this pass and will not next time either (engine._SLOTS_VERIFIED)
        if all(p.grad is not None and p.grad.data_ptr() == g.data_ptr() for p, g in zip(params, grads)):
            engine._SLOTS_VERIFIED.update(id(p) for p in params)
            state.verified.add(idx)
    group = state.group if state.group is not None else dist.group.WORLD
    world = dist.get_world_size(group)
    if world == 1:
        # nothing to reduce; a node that deferred its join is ordered in front of the optimiser by the
        # engine's end-of-backward callback (engine.Run.defer_side)
        fut = torch.futures.Future()
        fut.set_result(buf)
        return fut
    # The bucket's gradients are written by weight-gradient kernels on the engine's side stream, and
    # intermediate stage nodes no longer join that stream on the main one (engine.Run.defer_side): the
    # pre-division and the all-reduce are ordered behind BOTH streams on a third one, so the main stream's
    # data-gradient chain never waits for a weight gradient.
    side = engine.side_stream_of(buf.device) if buf.is_cuda else None
    if side is None:
        buf.div_(world)
        work = collective("ddp bucket %d all_reduce (parallel.bucket_hook; main_nce.py:172)" % idx,
                          lambda: dist.all_reduce(buf, group=group, async_op=True),
                          buf.numel() * buf.element_size(), buf.device)
    else:
        prep = _PREP.get(buf.device)
        if prep is None:
            prep = _PREP[buf.device] = torch.cuda.Stream(device=buf.device)
        prep.wait_stream(torch.cuda.current_stream(buf.device))
        # ... behind the weight gradients of THIS bucket only: every node records one event on the weight-
        # gradient stream when its last weight gradient has been enqueued (engine.side_events_for); waiting
        # for the whole stream would also wait for later stages' weight gradients that are already queued
        # (an empty list = no node of this bucket deferred its join: the main stream, which `prep` waited for
        # above, already orders those gradients)
        for ev in engine.side_events_for(params):
            prep.wait_event(ev)
        with torch.cuda.stream(prep):
            buf.div_(world)
            work = collective("ddp bucket %d all_reduce (parallel.bucket_hook; main_nce.py:172)" % idx,
                              lambda: dist.all_reduce(buf, group=group, async_op=True),
                              buf.numel() * buf.element_size(), buf.device)
    return work.get_future().then(lambda f: f.value()[0])


def install(module_types):
    """Make `gradient_as_bucket_view=True` the default for DDP wrappers around `module_types`."""
    if os.environ.get("COCLR_PATCH_DDP", "1") == "0" or _installed[0]:
        return False
    orig = _DDP.__init__

    def __init__(self, module, *args, **kwargs):
        ours = isinstance(module, module_types) and "gradient_as_bucket_view" not in kwargs and \
            len(args) <= _POS
        if ours:
            kwargs["gradient_as_bucket_view"] = True
        orig(self, module, *args, **kwargs)
        if ours and os.environ.get("COCLR_DDP_HOOK", "1") != "0":
            handle = self.register_forward_pre_hook(_register_lazily)
            self.__dict__["_coclr_hook_pending"] = (_HookState(self.process_group), handle)

    __init__.__wrapped__ = orig
    __init__.__doc__ = orig.__doc__
    _DDP.__init__ = __init__
    _installed[0] = True
    if os.environ.get("COCLR_QUIET", "0") != "1":
        print("coclr_amd: DistributedDataParallel defaults to gradient_as_bucket_view=True for "
              "InfoNCE/UberNCE/CoCLR modules and, unless the caller registers a communication hook of their "
              "own before the first forward, averages per bucket through one (COCLR_PATCH_DDP=0 / "
              "COCLR_DDP_HOOK=0 opt out)", file=sys.stderr)
    return True
