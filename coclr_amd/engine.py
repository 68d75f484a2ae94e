"""Layer-program executor for the conv backbones.

The reference lets ATen's autograd record ~255 tiny ops per encoder pass
(backbone/s3dg.py:211-217).  Here a backbone forward is one *run*: modules emit
kernel launches through this executor, which keeps its own reverse tape of
closures.  A single torch.autograd.Function (`EngineFn`) wraps the whole run,
so autograd / DDP only ever see "backbone(x, *params) -> feature map" and
receive all parameter gradients when the tape has been replayed.

What that buys on MI355X:
  * branches of an inception block write straight into channel slices of the
    block output (no torch.cat), gradients are read back from slices
  * conv -> BN statistics are produced by the conv epilogue, BN+ReLU is one
    streaming pass, its backward one reduce + one apply pass
  * data gradients of fan-out nodes accumulate in place (accumulate flag on the
    kernel) instead of separate add kernels
  * the launch sequence is static per input shape (hipGraph-capturable)
"""
import os
import weakref

import torch

from . import ops
from . import plan as _plan


_PACK_STORE = {}      # id(weight) -> (weakref, {use -> packed buffer})
# One launch re-lays every weight operand of a module's pass (recorded on its first pass) instead
# of ~80 five-microsecond launches threaded between the convolutions.
BATCH_PACK = os.environ.get("COCLR_BATCH_PACK", "1") != "0"
WGRAD_STREAM = os.environ.get("COCLR_WGRAD_STREAM", "1") != "0"
# Inception branches on their own streams: correct (GPU tier passes with it on) but SLOWER on
# MI355X -- 47.3 vs 43.3 ms/step: ~160 fork/join points per step cost more in cross-queue
# event latency than the overlap of the small branch kernels wins.  Off unless asked for.
# COCLR_LANES: "1" everywhere; "small" only in blocks on maps of 8x8 and below (stages 4 and 5, whose
# 10-40 us kernels leave most of the chip idle); "graph" only while the pass is being captured into a
# hipGraph (the forks become graph branches, no host-issued cross-queue events); "small+graph" both.
LANES_MODE = os.environ.get("COCLR_LANES", "0")
LANES = LANES_MODE != "0"


def lanes_for(run, dims):
    """Should the inception block on a map of extent `dims` run its branches on lane streams?"""
    if not LANES or run.device.type != "cuda":
        return False
    if LANES_MODE == "1":
        return True
    ok = True
    if "small" in LANES_MODE:
        ok = ok and dims[1] <= 8
    if "graph" in LANES_MODE:
        ok = ok and torch.cuda.is_current_stream_capturing()
    return ok
# Test instrumentation (tests/_decisions.py): when set, called as DECISION_PROBE("relu", bn, mask) for
# every differentiated BatchNorm+ReLU unit (mask = the unit's ReLU decisions, exactly the predicate the
# kernels evaluate: fma(y, scale, shift) (+ residual) > 0) and DECISION_PROBE("pool", None, argmax) for
# every differentiated max-pool, in emission order.  A gradient is a piecewise-smooth function of the
# inputs; the pieces are these decisions.  Never set by the product.
DECISION_PROBE = None


def _probe_relu(bn, y, scale, shift, residual=None):
    sh = (1, -1, 1, 1, 1)
    z = torch.addcmul(shift.double().view(sh), y.double(), scale.double().view(sh)).float()
    if residual is not None:
        z = z + residual
    DECISION_PROBE("relu", bn, z > 0)


_SIDE = {}
_PENDING_SIDE = {}    # device -> tensors read by weight-gradient kernels of nodes that deferred their join
DEFER_JOIN = os.environ.get("COCLR_DEFER_JOIN", "1") != "0"
DEFERRED = [0]        # nodes that left the stream un-joined (tests / bench read it)
# id(param) -> event on the weight-gradient stream behind the last kernel that writes the parameter's
# gradient in the current backward pass (one event per node that deferred its join), or None when the node
# joined the stream on the main one.  Read -- and consumed -- by the bucket hook (coclr_amd/parallel.py).
_SIDE_EVENTS = {}
_CALLBACK_QUEUED = {}  # device -> an end-of-backward join has been queued for the running backward pass


def _note(fn):
    """Run `fn` (a stream dependency, a table update: something a pass does beside its C-ABI calls) now and, when
    this thread is recording a launch plan (coclr_amd/plan.py), again at the same place of every replay."""
    fn()
    if _plan._ACTIVE is not None:
        rec = _plan.active()
        if rec is not None:
            rec.py(fn)


def side_stream_of(device):
    """The weight-gradient stream of `device`, if any work was ever put on it."""
    return _SIDE.get(device)


def side_events_for(params):
    """Events on the weight-gradient stream that the gradients of `params` are complete behind (duplicates
    removed).  Parameters without an entry were not written by a node that deferred its join -- the
    projection head (main stream), a node that joined, a graph replay (joins inside the graph): the main
    stream orders them."""
    evs = []
    for p in params:
        ev = _SIDE_EVENTS.pop(id(p), None)
        if ev is not None and not any(ev is e for e in evs):
            evs.append(ev)
    return evs


def _end_of_backward(device):
    """Autograd end-of-pass callback queued by the first node that deferred its join: if the node that ran
    LAST deferred as well (stage 1 frozen or not part of the graph), nobody has joined the weight-gradient
    stream -- do it here, in front of the optimiser, and release what its kernels read."""
    _CALLBACK_QUEUED[device] = False
    _PENDING_SIDE.pop(device, None)
    if device in _SIDE:        # unconditional: a join of a drained stream costs one event, a missing one a race
        torch.cuda.current_stream(device).wait_stream(_SIDE[device])


# The weight-gradient stream runs at HIGH priority (-1; HIP on MI355X has two levels, (0, -1)): it is the
# stream that finishes a backward pass last (the stem's weight gradients have nothing left to overlap with), so
# dispatching its kernels first shortens the tail: +0.5-0.9 % on the step, 5 of 5 alternating pairs on two boxes,
# also with one autograd node per stage (profiles/r05_wgrad_priority_ab.txt).  0 restores the default priority.
_SIDE_PRIORITY = int(os.environ.get("COCLR_WGRAD_PRIORITY", "-1"))
# weight-gradient closures per release window (Run.side_stream); 0 = hold everything until join_side
_SIDE_WINDOW = int(os.environ.get("COCLR_SIDE_WINDOW", "12"))
_LANES = {}


# Parameter -> the view of DistributedDataParallel's gradient bucket that holds its gradient
# (coclr_amd/parallel.py fills this from its communication hook).  A backward pass writes weight
# gradients STRAIGHT into those views and hands autograd a fresh alias of them: DDP then finds the
# gradient already in its bucket and skips its per-parameter copy (235 tiny launches per step on the
# critical stream).  Self-healing: a stale slot (buckets rebuilt, wrapper re-created) is just memory
# -- DDP's own alias check copies from it as from any other gradient -- and the hook refreshes it.
_GRAD_SLOTS = {}      # id(param) -> (weakref(param), bucket view)
# Slots the hook has SEEN DDP accept: a whole backward pass after which the bucket still lived at the
# published address and every gradient in it was an alias of its view.  Only then is it known that DDP
# launches nothing on the main stream that reads the gradient (its per-parameter copy out of a stale view
# after the one bucket rebuild does), which is what an un-joined weight-gradient stream needs (defer_side).
_SLOTS_VERIFIED = set()


def set_grad_slot(param, view):
    _SLOTS_VERIFIED.discard(id(param))
    if view is None:
        _GRAD_SLOTS.pop(id(param), None)
        return
    _GRAD_SLOTS[id(param)] = (weakref.ref(param), view)
    if len(_GRAD_SLOTS) > 8192:
        for k_ in [k_ for k_, v_ in _GRAD_SLOTS.items() if v_[0]() is None]:
            del _GRAD_SLOTS[k_]


_HEAD_SLOTS_OUT = set()   # parameters whose bucket view grad_out_for has handed out since the last forward


def new_pass(device):
    """Start of a differentiated forward: state that belongs to ONE forward/backward pair is reset here, because
    autograd runs no end-of-pass callback when a backward pass raises (OOM, a collective error): the
    end-of-backward join would otherwise never be queued again, and a bucket view handed out by a pass that died
    would stay 'taken'."""
    _CALLBACK_QUEUED[device] = False
    _HEAD_SLOTS_OUT.clear()


def grad_out_for(p):
    """Run.grad_out for code outside an engine run (the projection head, coclr_amd/model/pretrain.py): a
    fresh alias of p's DistributedDataParallel bucket view when there is one and `.grad` is unset -- the
    kernel then writes the gradient where DDP wants it and DDP copies nothing -- else new memory.  Once per
    parameter and pass: a second node over the same weights (the model run twice, one backward over both
    graphs) gets its own memory and autograd adds the two."""
    slot = _GRAD_SLOTS.get(id(p)) if _GRAD_SLOTS else None
    if slot is not None and slot[0]() is p and p.grad is None and id(p) not in _HEAD_SLOTS_OUT:
        v = slot[1]
        if v.device == p.device and v.shape == p.shape and v.dtype == p.dtype:
            _HEAD_SLOTS_OUT.add(id(p))
            return v.view_as(v)
    return torch.empty_like(p)


class _Lane:
    def __init__(self, run, idx):
        self.run, self.idx, self.ctx = run, idx, None

    def __enter__(self):
        run = self.run
        if run.cur_lane is not None or run._in_lane:
            raise RuntimeError("coclr_amd: lanes do not nest")
        run._in_lane = True
        if run.lanes_on:
            st, parent = run._lane_stream(self.idx)
            st.wait_stream(parent)
            run._parent = parent
            run._open.append(st)
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
            run.cur_lane = self.idx          # closures recorded inside replay on the lane's stream
        else:
            run.cur_lane = None
        self.entered = True
        return self

    def __exit__(self, *exc):
        self.run.cur_lane = None
        self.run._in_lane = False
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class PackPlan:
    """The weight re-layouts one module pass asks for (forward operands and, when the pass is
    differentiated, the transposed ones its backward uses), replayed as ONE kernel launch at the
    start of every later pass.  Weights may only change between passes (optimiser / momentum
    update), which is exactly when the launch scripts change them."""

    def __init__(self):
        self.requests = {}      # key -> (args, kwargs) of ops.conv_pack_weights
        self.table = None
        self.blockmap = None
        self.dirty = False

    def note(self, key, args, kw):
        if key not in self.requests:
            self.requests[key] = (args, kw)
            self.dirty = True

    def reset(self):
        self.requests.clear()
        self.table = self.blockmap = None
        self.dirty = False

    def build(self, device):
        rows, bmap = [], []
        self._ptrs = []
        for i, (args, kw) in enumerate(self.requests.values()):
            row, nb = ops.conv_pack_describe(*args, **kw)
            rows.append(row)
            bmap.extend((i, b) for b in range(nb))
            self._ptrs.append((args[0], args[0].data_ptr(), args[1], args[1].data_ptr()))
        self.table = torch.tensor(rows, dtype=torch.int64).to(device)
        self.blockmap = torch.tensor(bmap, dtype=torch.int32).to(device)
        self.dirty = False

    def launch(self, device):
        """Re-lay everything recorded so far; False when there is nothing to replay (first pass,
        or the table cannot be (re)built inside a stream capture)."""
        if not self.requests:
            return False
        capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if self.dirty or self.table is None:
            if capturing:
                return False
            self.build(device)
        # the table holds raw pointers: parameters re-created since (.to(), load with assign, ...)
        # invalidate it
        for w, wp, buf, bp in self._ptrs:
            if w.data_ptr() != wp or buf.data_ptr() != bp:
                self.reset()
                return False
        ops.conv_pack_batch(self.table, self.blockmap, list(self.requests.values()))
        return True


# While a differentiated pass is being captured into a hipGraph (GRAPH_QUERY below) the capture stream
# stands for the stream the graph will be replayed on: packed-operand buffers and re-layout plans
# recorded by the eager passes on that stream are the ones the captured kernels use.
_STREAM_ALIAS = {}


def _stream_key(device):
    if device.type != "cuda":
        return 0
    st = torch.cuda.current_stream(device).cuda_stream
    return _STREAM_ALIAS.get(st, st)


def _plan_for(module, save, device):
    if not BATCH_PACK:
        return None
    plans = module.__dict__.get("_coclr_packplans")
    if plans is None:
        plans = module.__dict__["_coclr_packplans"] = {}
    stream = _stream_key(device)
    key = (bool(save), stream, device)
    plan = plans.get(key)
    if plan is None:
        plan = plans[key] = PackPlan()
    return plan


class Val:
    """An activation: channels [c0, c0+C) of the dense NCDHW tensor `base`.

    `lazy` = (y, scale, shift, relu): the activation is relu(y*scale + shift) of a BatchNorm unit
    whose apply pass has NOT run yet.  A consumer that can apply the affine itself while reading
    (the max-pool kernels) takes `lazy` and the tensor is never written; anybody else calls
    `view()`, which runs the apply pass into `base` first."""
    __slots__ = ("base", "c0", "C", "lazy")

    def __init__(self, base, c0=0, C=None):
        self.base = base
        self.c0 = c0
        self.C = base.shape[1] - c0 if C is None else C
        self.lazy = None

    @property
    def whole(self):
        return self.c0 == 0 and self.C == self.base.shape[1]

    def view(self):
        v = self.base if self.whole else self.base[:, self.c0:self.c0 + self.C]
        if self.lazy is not None:
            if self.lazy == "consumed":
                raise RuntimeError("coclr_amd: activation was consumed in fused form (its tensor "
                                   "was never written) and cannot be read again")
            y, scale, shift, relu = self.lazy
            self.lazy = None
            ops.bn_act_apply(y, scale, shift, None, v, relu)
        return v

    @property
    def shape(self):
        b = self.base.shape
        return (b[0], self.C, b[2], b[3], b[4])


def _triple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v, v)


class Run:
    """One forward pass (and, if `save`, the tape to differentiate it)."""

    def __init__(self, device, save, need_input_grad=False):
        self.device = device
        self.save = save
        self.need_input_grad = need_input_grad
        # (closure, lane) in emission order.  Closures take the run as their ARGUMENT and must not
        # capture it: run -> tape -> closure -> run would be a cycle that only the cyclic gc frees,
        # one backbone stage per collection (each stage's tape holds the previous stage's output,
        # whose grad_fn owns that stage's run) -- GBs of activations parked behind gc's schedule.
        self.tape = []
        self.pooled = {}       # id(unit output) -> (pool geom, d(pool output), arg-max): see max_pool
        # BatchNorm backward sums formed by the data gradient that writes the unit's dz (FUSE_BN_REDUCE):
        self.bn_src = {}       # id(unit output) -> (y, scale, shift, mean, invstd, relu) of a training unit
        self.bn_parts = {}     # id(unit output) -> [(partial sums, slots per channel)] left by that writer
        self.grad_writes = {}  # id(activation) -> writers of its gradient so far (grad_target)
        self.cur_lane = None
        self.lanes_on = False  # set per inception block (lanes_for)
        self._in_lane = False
        self._parent = None
        self._open = []
        self.grads = {}        # id(base tensor) -> grad tensor
        self.param_grads = {}  # id(param) -> grad tensor
        self._slots_out = set()   # parameters whose bucket view has been handed out in this run
        self.no_grad_bases = set()
        self.out = None
        self.plan = None       # PackPlan of the module being run
        self.batched = False   # the plan's launch has re-laid its operands for this pass
        self._side_used = False
        self._side_keep = []   # tensors read by side-stream kernels, released at join_side()
        self._side_windows = []   # [(side-stream event, tensors enqueued before it)]
        self._side_calls = 0
        if device.type == "cuda" and device.index is not None and \
                device.index != torch.cuda.current_device():
            # kernels are enqueued on the CURRENT device's current stream (ops._stream)
            raise RuntimeError("coclr_amd: tensors live on %s but the current device is cuda:%d; "
                               "call torch.cuda.set_device first" % (device, torch.cuda.current_device()))

    def begin(self, module):
        """Attach the module's pack plan and replay it."""
        self.plan = _plan_for(module, self.save, self.device)
        if self.plan is not None:
            self.batched = self.plan.launch(self.device)

    # -- allocation helpers ------------------------------------------------------
    def empty(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def needs_grad(self, val):
        return self.save and id(val.base) not in self.no_grad_bases

    def grad_target(self, val):
        """(tensor to write d(val) into, accumulate?)"""
        if not val.whole:
            raise RuntimeError("coclr_amd: gradient into a channel slice is not supported")
        self.grad_writes[id(val.base)] = self.grad_writes.get(id(val.base), 0) + 1
        g = self.grads.get(id(val.base))
        if g is None:
            g = torch.empty_like(val.base)
            self.grads[id(val.base)] = g
            return g, False
        return g, True

    def grad_of(self, val):
        g = self.grads.get(id(val.base))
        if g is None:
            raise RuntimeError("coclr_amd: activation has no gradient (unused output?)")
        return g if val.whole else g[:, val.c0:val.c0 + val.C]

    def grad_out(self, p):
        """Tensor to write d(loss)/d(p) into: a fresh alias of p's bucket view when DDP has one and the
        caller's `.grad` is unset (zero_grad(set_to_none=True), torch's default: otherwise autograd
        would ADD this tensor to a `.grad` that aliases it), else new memory."""
        slot = _GRAD_SLOTS.get(id(p)) if _GRAD_SLOTS else None
        if slot is not None and slot[0]() is p and p.grad is None and id(p) not in self.param_grads \
                and id(p) not in self._slots_out:
            v = slot[1]
            if v.device == p.device and v.shape == p.shape and v.dtype == p.dtype:
                self._slots_out.add(id(p))
                return v.view_as(v)          # new tensor object: autograd adopts it without a copy
        return torch.empty_like(p)

    def add_param_grad(self, p, g):
        old = self.param_grads.get(id(p))
        if old is not None and _plan._ACTIVE is not None and _plan.active() is not None:
            _plan.active().taint("a parameter used twice in one pass (its gradients are added by an ATen launch)")
        self.param_grads[id(p)] = g if old is None else old.add_(g)

    # -- lanes: independent sub-graphs (inception branches) on their own streams ------
    def record(self, fn):
        self.tape.append((fn, self.cur_lane))

    def _lane_stream(self, lane):
        parent = torch.cuda.current_stream(self.device) if self.cur_lane is None else self._parent
        key = (self.device, parent.cuda_stream, lane)
        st = _LANES.get(key)
        if st is None:
            st = _LANES[key] = torch.cuda.Stream(device=self.device)
        return st, parent

    def lane(self, idx):
        """Context: kernels emitted inside run on lane stream `idx`, ordered after everything
        queued on the parent stream so far.  The caller closes the region with join_lanes()."""
        return _Lane(self, idx)

    def join_lanes(self):
        if self._open:
            cur = torch.cuda.current_stream(self.device)
            for st in self._open:
                cur.wait_stream(st)
            self._open = []

    # -- backward ----------------------------------------------------------------
    def backward(self, dout, defer_join=False):
        """defer_join: this node may leave the weight-gradient stream un-joined (see defer_side)."""
        out = self.out
        self.grads[id(out.base)] = dout.contiguous()
        # closures run in reverse emission order; closures of different lanes between two
        # main-stream closures are independent of each other, so each lane replays on its own
        # stream (entered after the main stream's work so far) and the next main-stream
        # closure joins them
        started = {}
        while self.tape:
            fn, lane = self.tape.pop()
            if lane is None:
                if started:
                    cur = torch.cuda.current_stream(self.device)
                    for st in started.values():
                        cur.wait_stream(st)
                    started = {}
                _run_closure(fn, self)
            else:
                st = started.get(lane)
                if st is None:
                    st, parent = self._lane_stream(lane)
                    st.wait_stream(parent)
                    started[lane] = st
                with torch.cuda.stream(st):
                    _run_closure(fn, self)
        if started:
            cur = torch.cuda.current_stream(self.device)
            for st in started.values():
                cur.wait_stream(st)
        if defer_join and self._side_used and self.param_grads and \
                all(k in self._slots_out and k in _SLOTS_VERIFIED for k in self.param_grads):
            self.defer_side()
        else:
            self.join_side()
        return self.grads

    # Weight gradients feed nothing inside the backward pass (only the optimiser), so they
    # run on a second stream: the dgrad -> BN-backward chain of the next unit proceeds on
    # the main stream while the wgrad of this unit fills whatever CUs it leaves idle.
    def side_stream(self, *inputs):
        """Stream for work that depends on `inputs` (tensors produced on the main stream) but
        that nothing later on the main stream waits for until `join_side`."""
        if not (WGRAD_STREAM and self.device.type == "cuda"):
            return None
        st = _SIDE.get(self.device)
        if st is None:
            st = _SIDE[self.device] = torch.cuda.Stream(device=self.device, priority=_SIDE_PRIORITY)
        cur = torch.cuda.current_stream(self.device)
        _note(lambda st=st, cur=cur: st.wait_stream(cur))
        # Everything the side kernels read (the gradient dy AND the saved activation x) stays
        # referenced until join_side(): once a closure is popped its tensors would otherwise go
        # back to the main stream's allocator pool and could be handed out again while a lagging
        # weight-gradient kernel still reads them.  Holding the references is cheaper on the host
        # than record_stream (no event queries at free time) and bounded by one backward pass.
        # ... in windows: every _SIDE_WINDOW calls an event on the side stream closes the window
        # of everything enqueued there so far; windows whose event has completed drop their
        # references (one event query per window, so the host cost stays negligible while the
        # peak is activations + a few windows of dy instead of activations + ALL dy).
        self._side_calls += 1
        # (never while a launch plan is recorded: a release justified by an event QUERY is a fact about this
        # pass's timing; the addresses it frees would be re-used by the plan at every replay, whose weight-
        # gradient stream may lag further behind)
        if _SIDE_WINDOW and self._side_calls % _SIDE_WINDOW == 0 and self._side_keep and \
                not torch.cuda.is_current_stream_capturing() and \
                not (_plan._ACTIVE is not None and _plan.active() is not None):
            ev = torch.cuda.Event()
            ev.record(st)
            self._side_windows.append((ev, self._side_keep))
            self._side_keep = []
            while self._side_windows and self._side_windows[0][0].query():
                self._side_windows.pop(0)
        self._side_keep.extend(inputs)
        self._side_used = True
        return st

    def join_side(self):
        pending = _PENDING_SIDE.get(self.device)
        if self._side_used or pending:
            dev, keys = self.device, tuple(self.param_grads)

            def join(dev=dev, keys=keys):
                _PENDING_SIDE.pop(dev, None)
                torch.cuda.current_stream(dev).wait_stream(_SIDE[dev])
                if _SIDE_EVENTS:
                    for k in keys:
                        _SIDE_EVENTS.pop(k, None)         # ordered by the main stream from here on
            _note(join)
            self._side_used = False
        elif _SIDE_EVENTS:
            for k in self.param_grads:
                _SIDE_EVENTS.pop(k, None)
        self._side_keep = []
        self._side_windows = []

    def defer_side(self):
        """Leave the weight-gradient stream un-joined at the end of an intermediate autograd node.

        With the backbone as one node per stage (world > 1) every node used to end with a join: the next
        stage's data-gradient chain then waited for this stage's weight gradients.  Nothing on the main
        stream needs them: when EVERY parameter gradient of the node was written into DistributedDataParallel's
        bucket views (grad_out) and those views are known to be DDP's current ones (_SLOTS_VERIFIED: out of a
        stale view DDP would copy on the main stream), autograd only adopts aliases (no kernel reads them)
        and the bucket hook (coclr_amd/parallel.py) makes the all-reduce wait for the weight-gradient stream
        itself.  The last
        node of the backward pass (stage 1) joins as before, which also orders everything in front of the
        optimiser.  What the side kernels read stays referenced until then."""
        keep = _PENDING_SIDE.setdefault(self.device, [])
        keep.extend(self._side_keep)
        for _, ts in self._side_windows:
            keep.extend(ts)
        self._side_keep = []
        self._side_windows = []
        self._side_used = False
        # one event behind this node's weight gradients: the bucket hook waits for the events of the
        # parameters in ITS bucket instead of for everything queued on the stream by then
        # ... and whoever runs last joins: the last node (stage 1) does it itself (join_side); if that one
        # deferred too, or never runs, the end-of-pass callback does
        dev, keys = self.device, tuple(self.param_grads)

        def publish(dev=dev, keys=keys):
            DEFERRED[0] += 1
            ev = torch.cuda.Event()
            ev.record(_SIDE[dev])
            for k in keys:
                _SIDE_EVENTS[k] = ev
            if not _CALLBACK_QUEUED.get(dev):
                _CALLBACK_QUEUED[dev] = True
                torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(dev))
        _note(publish)

    # -- weight packing ------------------------------------------------------------
    def _packed_buffer(self, owner, tag, n, zero):
        """Packed-operand buffer that lives with the weight tensor `owner` (one per use and per
        stream): re-packed every run, never re-allocated.  Safe because a given encoder always
        runs on the same stream, so the re-pack is ordered after the previous run's reads."""
        store = _PACK_STORE.get(id(owner))
        if store is None or store[0]() is not owner:
            store = _PACK_STORE[id(owner)] = (weakref.ref(owner), {})
            if len(_PACK_STORE) > 4096:
                for k_ in [k_ for k_, v_ in _PACK_STORE.items() if v_[0]() is None]:
                    del _PACK_STORE[k_]
        if self.device.type != "cuda":
            stream = 0
        elif self.cur_lane is not None and self._parent is not None:
            # inside a lane: the buffers (and the batch re-layout that fills them) belong to the parent
            # stream, which the lane stream waited for when it forked
            stream = _STREAM_ALIAS.get(self._parent.cuda_stream, self._parent.cuda_stream)
        else:
            stream = _stream_key(self.device)
        key = (tag, n, stream, self.device)
        buf = store[1].get(key)
        if buf is None:
            buf = (torch.zeros if zero else torch.empty)(n, dtype=torch.float32, device=self.device)
            store[1][key] = buf
        return buf

    def pack(self, w, transpose, kt_slice=None, taps=None, tap_base=0, tap_step=1, algo=0):
        """[Cout][Cin][taps] -> [taps][Cin'][Cout'] (or the dgrad operand).  kt_slice picks
        one temporal slice of the stencil; (taps, tap_base, tap_step) an arithmetic subset;
        algo=1 the Winograd-domain matrices of a (3,1,1) or (1,3,3) stencil."""
        cout, cin, kt, kh, kw = w.shape
        if algo >= 1 and taps is not None:
            # a 3- or 4-tap arithmetic subset of a temporal stencil (one phase of a strided data gradient):
            # 6 matrices of F(4,3) / 5 of F(2,4)
            vt = {3: 6, 4: 5}[taps]
            n = ops.conv_packed_size(cin, cout, vt, transpose)
            packed = self._packed_buffer(w, ("wino", bool(transpose), vt, tap_base, tap_step), n, False)
            self._relayout(w, packed, (cout, cin, vt, cin * kt * kh * kw, kt * kh * kw, tap_base,
                                       int(bool(transpose)) | 2, tap_step), {})
            return packed
        if algo >= 1:
            # (3,1,1): 4 matrices of F(2,3) (algo 1) or 6 of F(4,3) (algo 2); (1,3,3): 16 matrices of F(2x2,3x3)
            # (7,1,1)/2: the 4 + 5 polyphase matrices of the temporal stem conv
            vt = (6 if algo == 2 else 4) if kt == 3 else (9 if kt == 7 else 16)
            n = ops.conv_packed_size(cin, cout, vt, transpose)
            packed = self._packed_buffer(w, ("wino", bool(transpose), vt), n, False)
            self._relayout(w, packed, (cout, cin, vt, cin * kt * kh * kw, kt * kh * kw, 0,
                                       int(bool(transpose)) | 2, 1), {})
            return packed
        if taps is not None:
            base = tap_base
        elif kt_slice is None:
            taps, base = kt * kh * kw, 0
        else:
            taps, base = kh * kw, kt_slice * kh * kw
        n = ops.conv_packed_size(cin, cout, taps, transpose)
        packed = self._packed_buffer(w, (bool(transpose), taps, base, tap_step), n, False)
        self._relayout(w, packed, (cout, cin, taps, cin * kt * kh * kw, kt * kh * kw, base,
                                   int(bool(transpose)), tap_step), {})
        return packed

    def pack_concat(self, weights, transpose):
        """Pointwise weights [Cout_i][Cin] of convs sharing one input, as ONE packed operand over
        the concatenated output channels (forward) / reduction rows (data gradient)."""
        cin = weights[0].shape[1]
        ctot = sum(w.shape[0] for w in weights)
        n = ops.conv_packed_size(cin, ctot, 1, transpose)
        # zero padding is written once, when the buffer is created; every run rewrites exactly
        # the real sub-blocks
        packed = self._packed_buffer(weights[0], ("cat", bool(transpose), ctot), n, True)
        c0 = 0
        for w in weights:
            cout = w.shape[0]
            if transpose:
                self._relayout(w, packed, (cout, cin, 1, cin, 1, 0, 1, 1),
                               dict(row0=c0, rows_total=ctot, col0=0, cols_total=cin))
            else:
                self._relayout(w, packed, (cout, cin, 1, cin, 1, 0, 0, 1),
                               dict(row0=0, rows_total=cin, col0=c0, cols_total=ctot))
            c0 += cout
        return packed

    def _relayout(self, w, packed, args, kw):
        """ops.conv_pack_weights(w, packed, *args, **kw), unless this pass's batch launch has
        already done it; recorded for the batch launch of later passes."""
        plan = self.plan
        if plan is not None:
            key = (id(w), packed.data_ptr(), args, tuple(sorted(kw.items())))
            if self.batched and key in plan.requests:
                return
            plan.note(key, (w, packed) + args, kw)
        ops.conv_pack_weights(w, packed, *args, **kw)


# ---------------------------------------------------------------------------------
# conv (+ BatchNorm) (+ residual) (+ ReLU)
# ---------------------------------------------------------------------------------

_SLICED = {}


def _sliced_geoms(N, Cin, Cout, idim, k, s, p):
    """(5,7,7) stem as one (1,7,7) launch per temporal tap, all writing the full output."""
    key = (N, Cin, Cout, tuple(idim), k, s, p)
    g = _SLICED.get(key)
    if g is None:
        odim = ops.ConvGeom(N, Cin, Cout, idim, k, s, p).odim
        g = _SLICED[key] = [ops.ConvGeom(N, Cin, Cout, idim, (1, k[1], k[2]), s,
                                         (p[0] - t, p[1], p[2]), odim=odim) for t in range(k[0])]
    return g


def _is_plain_pointwise(k, s):
    return k == (1, 1, 1) and s != (1, 1, 1)


# ---------------------------------------------------------------------------------
# Units as coroutines: sibling units in lockstep, their launches fused pairwise
# ---------------------------------------------------------------------------------
# A conv unit (conv_bn_act) is written as a generator that YIELDS its main-stream launches as requests
#   ("conv", {...})    forward convolution or data gradient      -> (statistics buffer, slots per channel)
#   ("bn_fwd", {...})  BatchNorm finalize + apply (+ReLU)
#   ("bn_bwd", {...})  BatchNorm(+ReLU) backward
# instead of calling ops directly.  Driven alone (`_drive`) every request is executed at once: the launch
# sequence is exactly what direct calls would give.  Two INDEPENDENT units with the same structure -- the
# separable branch1 / branch2 tails of an inception block (backbone/s3dg.py:100-118) -- are driven in
# lockstep (`drive_pair`): requests of the same kind become ONE call of the library's multi entry points
# (coclr_conv3d_fwd_multi, coclr_bn_finalize_apply_multi, coclr_bn_act_backward_multi), which launch both
# problems as one grid where the kernels allow it.  The backward closures the two units record are
# generators as well and are paired on the tape, so the backward pass fuses the same way.  Weight gradients
# (side stream, nothing waits for them) are not requests.
PAIR_UNITS = os.environ.get("COCLR_PAIR_UNITS", "1") != "0"


def _exec(req):
    kind, c = req
    if kind == "conv":
        stats = nt = None
        if c.get("want_stats"):
            nt = c["geom"].ntiles()
            stats = torch.empty(2 * c["geom"].Cout * nt, dtype=torch.float32, device=c["y"].device)
        if c.get("bwd_bn") is not None or c.get("in_affine") is not None:
            c["stats"] = stats
            ops.conv_fwd_multi([c])
        else:
            ops.conv_fwd(c["geom"], c["x"], c["w"], c["y"], stats=stats, n_index=c.get("n_index"),
                         accumulate=c.get("accumulate", False))
        return stats, nt
    if kind == "bn_fwd":           # c: list of units
        if len(c) > 1:
            ops.bn_finalize_apply_multi(c)
            return None
        u = c[0]
        gamma, beta, rm, rv, nbt, momentum, eps = u["bn"]
        mean, invstd, scale, shift = u["small"]
        ops.bn_finalize_apply(u["stats"], u["C"], u["ntiles"], u["count"], gamma, beta, rm, rv, nbt, momentum,
                              eps, mean, invstd, scale, shift, u["y"], u["z"], u["relu"], c0=u.get("c0", 0),
                              c_total=u.get("c_total"))
        return None
    if kind == "bn_bwd":           # c: list of units
        if len(c) > 1:
            ops.bn_act_backward_multi(c)
            return None
        u = c[0]
        if u.get("partials"):
            ops.bn_act_backward_multi(c)
            return None
        ops.bn_act_backward(u["dz"], u["y"], None, u["scale"], u["shift"], u["mean"], u["invstd"], u["sums"],
                            u["dy"], None, u["dgamma"], u["dbeta"], u["relu"], u["training"])
        return None
    raise RuntimeError("coclr_amd: unknown launch request %r" % (kind,))


def _exec_pair(ra, rb):
    kind = ra[0]
    if kind != rb[0]:
        return _exec(ra), _exec(rb)
    ca, cb = ra[1], rb[1]
    if kind == "conv":
        res = [(None, None), (None, None)]
        for i, c in enumerate((ca, cb)):
            if c.get("want_stats"):
                nt = c["geom"].ntiles()
                c["stats"] = torch.empty(2 * c["geom"].Cout * nt, dtype=torch.float32, device=c["y"].device)
                res[i] = (c["stats"], nt)
        ops.conv_fwd_multi([ca, cb])
        return res[0], res[1]
    if kind == "bn_fwd":
        ops.bn_finalize_apply_multi(list(ca) + list(cb))
        return None, None
    if kind == "bn_bwd":
        ops.bn_act_backward_multi(list(ca) + list(cb))
        return None, None
    return _exec(ra), _exec(rb)


def _drive(gen):
    """Run a unit coroutine alone: every request is executed when it is made."""
    try:
        req = next(gen)
        while True:
            req = gen.send(_exec(req))
    except StopIteration as e:
        return e.value


def _run_closure(fn, run):
    """A tape entry: a plain function, or a generator function whose requests are executed one by one."""
    r = fn(run)
    if r is not None and hasattr(r, "send"):
        _drive(r)


class _PairedBackward:
    """Tape entry for two units emitted in lockstep: their backward coroutines run in lockstep too."""
    __slots__ = ("fa", "fb")

    def __init__(self, fa, fb):
        self.fa, self.fb = fa, fb

    def __call__(self, run):
        ga, gb = self.fa(run), self.fb(run)
        ga = ga if ga is not None and hasattr(ga, "send") else None
        gb = gb if gb is not None and hasattr(gb, "send") else None
        if ga is not None and gb is not None:
            drive_pair(run, ga, gb)
        else:
            for g in (ga, gb):
                if g is not None:
                    _drive(g)


def drive_pair(run, ga, gb):
    """Run two unit coroutines in lockstep, fusing requests of the same kind; returns their values.
    Closures they record on the run's tape are collected per unit and appended as paired entries."""
    main_tape = run.tape
    tapes = ([], [])
    gens = [ga, gb]
    reqs, vals, done = [None, None], [None, None], [False, False]

    def advance(i, first, value=None):
        run.tape = tapes[i]
        try:
            reqs[i] = next(gens[i]) if first else gens[i].send(value)
        except StopIteration as e:
            done[i], vals[i], reqs[i] = True, e.value, None
        finally:
            run.tape = main_tape

    advance(0, True)
    advance(1, True)
    while not (done[0] and done[1]):
        if not done[0] and not done[1]:
            ra, rb = _exec_pair(reqs[0], reqs[1])
            advance(0, False, ra)
            advance(1, False, rb)
        else:
            i = 0 if not done[0] else 1
            advance(i, False, _exec(reqs[i]))
    ta, tb = tapes
    for i in range(max(len(ta), len(tb))):
        if i < len(ta) and i < len(tb) and ta[i][1] is None and tb[i][1] is None:
            main_tape.append((_PairedBackward(ta[i][0], tb[i][0]), None))
        else:
            if i < len(ta):
                main_tape.append(ta[i])
            if i < len(tb):
                main_tape.append(tb[i])
    return vals[0], vals[1]


def conv_bn_act(run, x, conv, bn, relu=True, out=None, residual=None, n_index=None):
    """One conv unit of the backbone, executed at once (see conv_bn_act_gen)."""
    return _drive(conv_bn_act_gen(run, x, conv, bn, relu=relu, out=out, residual=residual, n_index=n_index))


def conv_bn_act_gen(run, x, conv, bn, relu=True, out=None, residual=None, n_index=None):
    """One conv unit of the backbone, as a coroutine of launch requests (see above).

    x: Val.  conv: nn.Conv3d (bias-free) holding the weight.  bn: nn.BatchNorm3d.
    out: optional Val (channel slice of a concat buffer) to receive the activation.
    residual: optional Val added before the ReLU (ResNet bottleneck tail).
    n_index: optional int64 tensor -- sample n of the conv input is x[n_index[n]].
    Mirrors BasicConv3d.forward / the halves of STConv3d.forward
    (backbone/s3dg.py:24-28,58-65) and the conv/bn/relu triplets of
    backbone/resnet_2d3d.py:67-86.
    """
    w = conv.weight
    k = tuple(w.shape[2:])
    s, p = _triple(conv.stride), _triple(conv.padding)
    if conv.bias is not None or _triple(conv.dilation) != (1, 1, 1) or conv.groups != 1:
        raise NotImplementedError("coclr_amd: backbone convs are bias-free, dense, undilated")
    if _is_plain_pointwise(k, s):
        # strided 1x1x1 (ResNet downsample): subsample first, then a dense pointwise conv
        x = subsample(run, x, s)
        s = (1, 1, 1)
    N = n_index.shape[0] if n_index is not None else x.shape[0]
    Cin, idim = x.shape[1], x.shape[2:]
    Cout = w.shape[0]
    if w.shape[1] != Cin:
        raise ValueError("coclr_amd: conv expects %d input channels, got %d" % (w.shape[1], Cin))
    training = bn.training or (bn.running_mean is None)

    sliced = k[0] > 3 and k[1] > 1          # (5,7,7) stem: one launch per temporal tap
    if sliced:
        geoms = _sliced_geoms(N, Cin, Cout, idim, k, s, p)
    else:
        geoms = [ops.conv_geom(N, Cin, Cout, idim, k, s, p)]
        if n_index is not None and geoms[0].algo and k[1] > 1:
            geoms = [ops.ConvGeom(N, Cin, Cout, idim, k, s, p)]    # the gather lives in the direct kernel
    # The input is a BatchNorm unit whose apply pass has not run (Val.lazy) and this convolution's kernel can apply
    # relu(y * scale + shift) while it reads (the polyphase temporal stem conv): the normalised tensor -- 1 GB at
    # B = 32 behind Conv_1a.conv1 -- is never written or re-read.  Only in passes that keep no tape: the weight
    # gradient of this unit reads the normalised tensor.
    in_affine = None
    if IN_AFFINE and x.lazy is not None and x.lazy != "consumed" and x.whole and not run.save and \
            n_index is None and not sliced and want_in_affine(geoms[0], training or run.save):
        ysrc, sc_in, sh_in, relu_in = x.lazy
        x.lazy = "consumed"
        xv, in_affine = ysrc, (sc_in, sh_in, relu_in)
    else:
        xv = x.view()
    odim = geoms[0].odim
    want_y = training or run.save or residual is not None

    small = run.empty(4, Cout)       # mean, invstd, scale, shift
    mean, invstd, scale, shift = small[0], small[1], small[2], small[3]
    lazy_ok = out is None         # a private output tensor: nobody else writes or reads it yet
    if out is None:
        out = Val(run.empty(N, Cout, *odim))
    zv = out.base if out.whole else out.view()
    y = None
    if want_y:
        y = run.empty(N, Cout, *odim)
        stats = ntiles = None
        for t, g in enumerate(geoms):
            last = t == len(geoms) - 1
            res = yield ("conv", dict(geom=g, x=xv, w=run.pack(w, False, t if sliced else None, algo=g.algo),
                                      y=y, want_stats=training and last, n_index=n_index, accumulate=t > 0,
                                      in_affine=in_affine))
            if training and last:
                stats, ntiles = res
        count = N * odim[0] * odim[1] * odim[2]
        lazy = LAZY_APPLY and residual is None and lazy_ok and count > ops.SMALL_CHANNEL
        if training:
            if bn.momentum is None:
                raise NotImplementedError("coclr_amd: cumulative-average BatchNorm momentum")
            if residual is None and not lazy:
                # statistics + apply in one call (a single launch for the small late-stage layers)
                yield ("bn_fwd", [dict(stats=stats, C=Cout, ntiles=ntiles, count=count,
                                       bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           bn.num_batches_tracked, float(bn.momentum), float(bn.eps)),
                                       small=(mean, invstd, scale, shift), y=y, z=zv, relu=relu)])
            else:
                ops.bn_finalize(stats, Cout, ntiles, count, bn.weight, bn.bias,
                                bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                float(bn.momentum), float(bn.eps), mean, invstd, scale, shift)
        else:
            ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                               Cout, mean, invstd, scale, shift)
        if lazy:
            out.lazy = (y, scale, shift, relu)       # applied by the consumer (see Val)
        elif not (training and residual is None):
            ops.bn_act_apply(y, scale, shift, residual.view() if residual is not None else None, zv,
                             relu)
    else:
        # inference with frozen statistics: fold BN+ReLU into the conv epilogue
        ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                           Cout, mean, invstd, scale, shift)
        for t, g in enumerate(geoms):
            last = t == len(geoms) - 1
            ops.conv_fwd(g, xv, run.pack(w, False, t if sliced else None, algo=g.algo), zv,
                         ep_scale=scale if last else None, ep_shift=shift if last else None,
                         relu=relu and last, n_index=n_index, accumulate=t > 0)

    if run.save and DECISION_PROBE is not None and relu:
        _probe_relu(bn, y, scale, shift, residual.view() if residual is not None else None)

    if run.save:
        x_needs = run.needs_grad(x)
        if n_index is not None and x_needs:
            raise NotImplementedError("coclr_amd: gathered conv input cannot require grad")
        if FUSE_BN_REDUCE and training and residual is None and out.whole and \
                N * odim[0] * odim[1] * odim[2] > ops.SMALL_CHANNEL:
            # whoever writes d(out) alone and in one piece may form this unit's backward sums on the way
            run.bn_src[id(out.base)] = (y, scale, shift, mean, invstd, relu)

        def backward(run):
            dgb = (run.grad_out(bn.weight), run.grad_out(bn.bias))
            sums = run.empty(ops.bn_backward_workspace(N, Cout), dtype=torch.float64)
            pooled = run.pooled.pop(id(out.base), None)
            # Nobody needs this unit's data gradient (the stem over the clip) and its weight-gradient kernel can
            # apply BatchNorm's backward while it loads: d(conv output) -- 1 GB at B = 32 -- is never written.
            fused = WGRAD_BN and pooled is None and residual is None and not x_needs and not sliced and \
                w.requires_grad and id(out.base) not in run.bn_parts and geoms[0].wgrad_bn_ok()
            dy = None if fused else torch.empty_like(y)
            if fused:
                coef = run.empty(5, Cout)
                dz = run.grad_of(out)
                ops.bn_act_backward_coeffs(dz, y, scale, shift, mean, invstd, sums, coef, dgb[0], dgb[1], relu,
                                           training)
            elif pooled is not None:
                # the unit's only reader was a max-pool that applied BN+ReLU itself: its backward left
                # (geometry, d(pool output), arg-max) here and d(activation) is never materialised
                ops.bn_act_backward_pooled(pooled[0], pooled[1], pooled[2], y, scale, shift, mean,
                                           invstd, sums, dy, dgb[0], dgb[1], relu, training)
            elif residual is None:
                parts = run.bn_parts.pop(id(out.base), None)
                if parts is not None and run.grad_writes.get(id(out.base), 0) != 1:
                    parts = None          # somebody else wrote d(out) too: the sums are stale
                yield ("bn_bwd", [dict(dz=run.grad_of(out), y=y, scale=scale, shift=shift, mean=mean,
                                       invstd=invstd, sums=sums, dy=dy, dgamma=dgb[0], dbeta=dgb[1], relu=relu,
                                       training=training, partials=parts)])
            else:
                dz = run.grad_of(out)
                dres = None
                dres_acc = False
                if run.needs_grad(residual):
                    dres, dres_acc = run.grad_target(residual)
                ops.bn_act_backward(dz, y, zv, scale, shift, mean, invstd, sums, dy, dres, dgb[0], dgb[1],
                                    relu, training, dres_acc)
            if bn.weight.requires_grad:
                run.add_param_grad(bn.weight, dgb[0])
            if bn.bias.requires_grad:
                run.add_param_grad(bn.bias, dgb[1])
            if fused:
                with torch.cuda.stream(run.side_stream(dz, y, coef, xv)):
                    dw = run.grad_out(w)
                    kk = w.shape[2] * w.shape[3] * w.shape[4]
                    ops.conv_wgrad_bn(geoms[0], xv, dz, y, coef, relu, dw, run.empty(geoms[0].wgrad_workspace()),
                                      Cin * kk, kk)
                run.add_param_grad(w, dw)
            elif w.requires_grad:
                with torch.cuda.stream(run.side_stream(dy, xv)):
                    dw = run.grad_out(w)
                    kk = w.shape[2] * w.shape[3] * w.shape[4]
                    for t, g in enumerate(geoms):
                        ws = run.empty(g.wgrad_workspace())
                        ops.conv_wgrad(g, xv, dy, dw, ws, Cin * kk, kk,
                                       t * k[1] * k[2] if sliced else 0)
                run.add_param_grad(w, dw)
            if x_needs:
                if sliced:
                    raise NotImplementedError("coclr_amd: dgrad of the sliced stem conv")
                dx, acc = run.grad_target(x)
                # dx is the dz of the unit that produced x: when this launch is its first writer (and, checked
                # by that unit, its only one) the kernel forms that unit's BatchNorm backward sums as well
                src = run.bn_src.get(id(x.base)) if (not acc and x.whole) else None
                phases = geoms[0].dgrad_phases()
                if phases is not None:
                    # strided conv: one dense stride-1 correlation per residue class of dX
                    if src is not None and not (len(phases) <= 2 and all(ph[0].bwd_sums_ok() for ph in phases)):
                        src = None
                    for pg, k0, nk, step in phases:
                        wp = run.pack(w, True, taps=nk, tap_base=k0, tap_step=step, algo=pg.algo)
                        if src is not None:
                            nt = pg.ntiles()
                            st = torch.empty(2 * pg.Cout * nt, dtype=torch.float32, device=dx.device)
                            ops.conv_fwd_multi([dict(geom=pg, x=dy, w=wp, y=dx, stats=st, bwd_bn=src)])
                            run.bn_parts.setdefault(id(x.base), []).append((st, nt))
                        else:
                            ops.conv_fwd(pg, dy, wp, dx, accumulate=acc)
                else:
                    dg = geoms[0].dgrad()
                    if src is not None and not dg.bwd_sums_ok():
                        src = None
                    res = yield ("conv", dict(geom=dg, x=dy, w=run.pack(w, True, algo=dg.algo), y=dx,
                                              accumulate=acc, want_stats=src is not None, bwd_bn=src))
                    if src is not None:
                        run.bn_parts.setdefault(id(x.base), []).append(res)

        run.record(backward)
    elif y is not None:
        del y
    return out


LAZY_APPLY = os.environ.get("COCLR_LAZY_APPLY", "1") != "0"
# BatchNorm's backward apply pass inside the weight gradient that is its only reader (the (1,7,7) stem)
WGRAD_BN = os.environ.get("COCLR_WGRAD_BN", "1") != "0"
# BatchNorm + ReLU of a unit applied by the CONVOLUTION that consumes it (gradient-free passes, kernels that can)
IN_AFFINE = os.environ.get("COCLR_IN_AFFINE", "1") != "0"


def want_in_affine(geom, keeps_y):
    """Can (and should) the kernel of `geom` apply the producing unit's BatchNorm + ReLU while it reads?  The
    polyphase temporal stem conv, in the form that writes y (train-mode statistics or a kept tape)."""
    return keeps_y and geom.k == (7, 1, 1) and geom.algo == 1
# BatchNorm backward sums in the epilogue of the data gradient that writes the unit's dz (single writer):
# one read of y there instead of the reduction pass over dz and y.  OPT-IN: measured on an MI355X at B=32
# (tools/r04_flaky.sh, four alternating pairs) it is worth nothing -- 1030.9 vs 1031.1 clips/s -- because
# the HBM-bound reduction passes it removes ran under the MFMA-bound weight gradients of the other stream,
# while the extra loads lengthen MFMA-bound kernels on the critical one.
FUSE_BN_REDUCE = os.environ.get("COCLR_FUSE_BN_REDUCE", "0") != "0"
POOLED_BACKWARD = os.environ.get("COCLR_POOLED_BACKWARD", "1") != "0"
FUSE_POINTWISE = True     # debugging switch: False runs the units of a group one by one


def pointwise_group(run, x, units):
    """pointwise_group_gen, executed at once."""
    return _drive(pointwise_group_gen(run, x, units))


def pointwise_group_gen(run, x, units):
    """Several 1x1x1 conv+BN+ReLU units that read the SAME input -- the heads of an inception
    block, branch0 / branch1[0] / branch2[0] (backbone/s3dg.py:97-104,119-123) -- executed as
    one convolution over their concatenated output channels: one pass over x instead of
    three, one weight-gradient GEMM, and one data-gradient GEMM that writes dX once
    instead of three read-modify-write passes.

    units: [(conv, bn, out Val or None)]; returns the activations as Vals.
    Falls back to separate launches when statistics are frozen and nothing is saved
    (BN+ReLU then fold into each conv's epilogue)."""
    training = all(bn.training or bn.running_mean is None for _, bn, _ in units)
    if not (training or run.save) or len(units) == 1 or not FUSE_POINTWISE:
        res = []
        for conv, bn, out in units:
            res.append((yield from conv_bn_act_gen(run, x, conv, bn, relu=True, out=out)))
        return res
    for conv, bn, _ in units:
        if tuple(conv.weight.shape[2:]) != (1, 1, 1) or _triple(conv.stride) != (1, 1, 1) or \
                _triple(conv.padding) != (0, 0, 0) or conv.bias is not None or \
                (bn.training or bn.running_mean is None) != training:
            raise NotImplementedError("coclr_amd: pointwise_group takes plain 1x1x1 conv units")
    N, Cin, idim = x.shape[0], x.shape[1], x.shape[2:]
    widths = [conv.weight.shape[0] for conv, _, _ in units]
    Ccat = sum(widths)
    weights = [conv.weight for conv, _, _ in units]
    geom = ops.conv_geom(N, Cin, Ccat, idim, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    xv = x.view()
    y = run.empty(N, Ccat, *idim)
    ntiles = geom.ntiles()
    stats, _ = yield ("conv", dict(geom=geom, x=xv, w=run.pack_concat(weights, False), y=y,
                                   want_stats=training))
    count = N * idim[0] * idim[1] * idim[2]
    outs, saved = [], []
    c0 = 0
    fwd_units = []
    for (conv, bn, out), C_ in zip(units, widths):
        small = run.empty(4, C_)
        mean, invstd, scale, shift = small[0], small[1], small[2], small[3]
        if training:
            if bn.momentum is None:
                raise NotImplementedError("coclr_amd: cumulative-average BatchNorm momentum")
            if out is None:
                out = Val(run.empty(N, C_, *idim))
            fwd_units.append(dict(stats=stats, C=C_, ntiles=ntiles, count=count,
                                  bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                      bn.num_batches_tracked, float(bn.momentum), float(bn.eps)),
                                  small=(mean, invstd, scale, shift), y=y[:, c0:c0 + C_], z=out.view(),
                                  relu=True, c0=c0, c_total=Ccat))
        else:
            ops.bn_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                               C_, mean, invstd, scale, shift)
            if out is None:
                out = Val(run.empty(N, C_, *idim))
            ops.bn_act_apply(y[:, c0:c0 + C_], scale, shift, None, out.view(), True)
        outs.append(out)
        saved.append((c0, C_, mean, invstd, scale, shift))
        c0 += C_
    if fwd_units:
        # the heads' BatchNorm units in one call (one launch on the 8x8x8 / 4x4x4 maps)
        yield ("bn_fwd", fwd_units)
    if run.save and DECISION_PROBE is not None:
        for (conv, bn, _), (c0_, C_, mean, invstd, scale, shift) in zip(units, saved):
            _probe_relu(bn, y[:, c0_:c0_ + C_], scale, shift)

    if run.save:
        x_needs = run.needs_grad(x)

        def backward(run):
            dy = torch.empty_like(y)
            bwd_units, dgbs = [], []
            for (conv, bn, _), out, (c0, C_, mean, invstd, scale, shift) in zip(units, outs, saved):
                dgb = (run.grad_out(bn.weight), run.grad_out(bn.bias))
                sums = run.empty(ops.bn_backward_workspace(N, C_), dtype=torch.float64)
                bwd_units.append(dict(dz=run.grad_of(out), y=y[:, c0:c0 + C_], scale=scale, shift=shift,
                                      mean=mean, invstd=invstd, sums=sums, dy=dy[:, c0:c0 + C_],
                                      dgamma=dgb[0], dbeta=dgb[1], relu=True, training=training))
                dgbs.append(dgb)
            yield ("bn_bwd", bwd_units)
            for (conv, bn, _), dgb in zip(units, dgbs):
                if bn.weight.requires_grad:
                    run.add_param_grad(bn.weight, dgb[0])
                if bn.bias.requires_grad:
                    run.add_param_grad(bn.bias, dgb[1])
            if any(w.requires_grad for w in weights):
                with torch.cuda.stream(run.side_stream(dy, xv)):
                    # one GEMM, delivered to each head's own gradient tensor
                    ws = run.empty(geom.wgrad_workspace())
                    if len(weights) <= 4:
                        dws = [run.grad_out(w) for w in weights]
                        ops.conv_wgrad(geom, xv, dy, dws, ws, Cin, 1, 0)
                    else:
                        dw = run.empty(Ccat, Cin, 1, 1, 1)
                        ops.conv_wgrad(geom, xv, dy, dw, ws, Cin, 1, 0)
                        dws = list(torch.split(dw, widths))
                for w, dw in zip(weights, dws):
                    if w.requires_grad:
                        run.add_param_grad(w, dw)
            if x_needs:
                dx, acc = run.grad_target(x)
                yield ("conv", dict(geom=geom.dgrad(), x=dy, w=run.pack_concat(weights, True), y=dx,
                                    accumulate=acc))

        run.record(backward)
    return outs


# ---------------------------------------------------------------------------------
# pooling
# ---------------------------------------------------------------------------------

def max_pool(run, x, kernel, stride, padding):
    """nn.MaxPool3d (backbone/s3dg.py:105,151,162,173,190; resnet_2d3d.py:141)."""
    N, Cc = x.shape[0], x.shape[1]
    g = ops.pool_geom(N, Cc, x.shape[2:], _triple(kernel), _triple(stride), _triple(padding))
    y = run.empty(N, Cc, *g.odim)
    need = run.needs_grad(x)
    idx = run.empty(N, Cc, *g.odim, dtype=torch.int32) if need else None
    fused = False
    if x.lazy is not None and x.whole:
        # BatchNorm + ReLU of the producing unit applied while the pool reads its raw output: the
        # normalised tensor is never written (its backward recomputes the ReLU mask from y)
        ysrc, scale, shift, relu = x.lazy
        x.lazy = "consumed"
        ops.maxpool_fwd(g, ysrc, y, idx, in_scale=scale, in_shift=shift, in_relu=relu)
        # ... and nobody else can read it, so the unit's backward can take the pool's gradient in
        # scattered form (conv_bn_act.backward -> ops.bn_act_backward_pooled)
        fused = POOLED_BACKWARD and ops.pooled_backward_fits(g)
    else:
        ops.maxpool_fwd(g, x.view(), y, idx)
    out = Val(y)
    if need and DECISION_PROBE is not None and tuple(g.k) != (1, 1, 1):     # a 1x1x1 window decides nothing
        DECISION_PROBE("pool", None, idx)
    if need:
        xid = id(x.base)

        def backward(run):
            if fused:
                run.pooled[xid] = (g, run.grad_of(out), idx)
                return
            dx, acc = run.grad_target(x)
            ops.maxpool_bwd(g, run.grad_of(out), idx, dx, accumulate=acc)
        run.record(backward)
    elif run.save:
        run.no_grad_bases.add(id(y))
    return out


def subsample(run, x, stride):
    """x[:, :, ::st, ::sh, ::sw] as a dense tensor (kernel-1 max pool)."""
    return max_pool(run, x, (1, 1, 1), stride, (0, 0, 0))


# ---------------------------------------------------------------------------------
# S3D-G feature gating (backbone/s3dg.py:68-78).  Not used by any benchmarked
# configuration; kept functional with device-side torch ops.
# ---------------------------------------------------------------------------------

def self_gating(run, x, fc, out=None):
    """SelfGating.forward (backbone/s3dg.py:73-78): out = x * sigmoid(fc(mean_{T,H,W} x))."""
    xv = x.view()
    N, C_ = xv.shape[0], xv.shape[1]
    S = xv.shape[2] * xv.shape[3] * xv.shape[4]
    if not xv.is_contiguous():
        raise NotImplementedError("coclr_amd: self gating expects a dense input")
    a5 = run.empty(N, C_, 1, 1, 1)
    ops.global_avgpool_fwd(xv, a5)
    a = a5.view(N, C_)
    s_ = run.empty(N, C_)
    W_ = fc.weight                                 # [C][C], s = a @ W^T + b
    ops.gemm(a, C_, 1, W_, 1, C_, s_, C_, fc.bias, N, C_, C_)
    wgt = run.empty(N, C_)
    ops.sigmoid_fwd(s_, wgt)
    if out is None:
        out = Val(torch.empty_like(xv))
    ops.plane_scale(xv, wgt, None, out.view())
    if run.save:
        def backward(run):
            dout = run.grad_of(out)
            dwgt = run.empty(N, C_)
            ops.plane_dot(dout, xv, dwgt)
            ds = run.empty(N, C_)
            ops.sigmoid_bwd(dwgt, wgt, ds)
            if fc.weight.requires_grad:
                dW = torch.empty_like(W_)          # dW[o][i] = sum_n ds[n][o] a[n][i]
                ops.gemm(ds, 1, C_, a, C_, 1, dW, C_, None, C_, C_, N)
                db = run.empty(C_)
                ops.colsum(ds, db)
                run.add_param_grad(fc.weight, dW)
                run.add_param_grad(fc.bias, db)
            if run.needs_grad(x):
                back = run.empty(N, C_)            # (ds @ W) / S, added to every element of a plane
                ops.gemm(ds, C_, 1, W_, C_, 1, back, C_, None, N, C_, C_, alpha=1.0 / S)
                g, acc = run.grad_target(x)
                ops.plane_scale(dout, wgt, back, g, accumulate=acc)
        run.record(backward)
    return out


# ---------------------------------------------------------------------------------
# autograd bridge
# ---------------------------------------------------------------------------------

class EngineFn(torch.autograd.Function):
    """backbone(x, *params) as ONE autograd node."""

    @staticmethod
    def forward(ctx, module, kwargs, x, *params):
        need_dx = x.requires_grad
        new_pass(x.device)
        run = Run(x.device, save=True, need_input_grad=need_dx)
        xin = Val(x if _dense5(x) else x.contiguous())
        if not need_dx:
            run.no_grad_bases.add(id(xin.base))
        run.begin(module)
        run.out = module._emit(run, xin, **kwargs)
        ctx.run = run
        ctx.xin = xin
        ctx.params = params
        ctx.need_dx = need_dx
        ctx.module_defer = getattr(module, "__dict__", {}).get("_coclr_defer_join", False)
        out = run.out.view()
        # A FRESH tensor object: autograd stamps what we return with grad_fn, and grad_fn -> ctx
        # -> run -> tape closures -> run.out.  Returning run.out's own tensor would close that
        # into a reference cycle (tensor -> grad_fn -> ... -> same tensor) that Python's gc cannot
        # see through, so a forward that is never followed by backward (main_coclr.py:403 skips
        # loss.backward() until the queue is full) would leak its whole activation tape.
        return out.detach() if out.is_contiguous() else out.contiguous()

    @staticmethod
    def backward(ctx, dout):
        run = ctx.run
        if run is None:
            raise RuntimeError("coclr_amd: backbone backward called twice (graph not retained)")
        ctx.run = None
        run.backward(dout, defer_join=DEFER_JOIN and bool(ctx.module_defer))
        dx = run.grads.pop(id(ctx.xin.base), None) if ctx.need_dx else None
        # Hand the gradients over WITHOUT keeping a reference: AccumulateGrad takes a freshly
        # produced gradient as .grad only when nobody else holds it, and clones it otherwise --
        # 235 extra copy launches per step when the run's tables still referenced them.
        grads = tuple(run.param_grads.pop(id(p), None) if p.requires_grad else None
                      for p in ctx.params)
        run.param_grads.clear()
        run.grads.clear()
        return (None, None, dx) + grads


# ---------------------------------------------------------------------------------
# hipGraph replay of a DIFFERENTIATED pass (the query encoder's forward and backward)
# ---------------------------------------------------------------------------------
# The launch sequence of a module pass is static per input shape: ~330 launches forward, ~600
# backward, ~18 ms of host work per training step between them (bench.py `host_floor_ms_per_step`).
# With COCLR_GRAPH_QUERY=1 a module that has been run eagerly a few times with an unchanged signature is
# captured -- forward and, at the first backward after that, its tape -- into two hipGraphs that share
# one private memory pool, and every later step is: copy the input into the graph's static buffer,
# replay, hand autograd fresh aliases of the static outputs / gradients.  Gradients that live in
# DistributedDataParallel's buckets (grad_out) are written there by the captured kernels as well.
# Same kernels, same order, same operands as the eager pass: results are bit-identical
# (tests/test_gpu_model.py::test_graphed_query_encoder_matches_eager).
# COCLR_GRAPH_QUERY=late captures only modules flagged `_coclr_graph_late` -- the 8x8x8 / 4x4x4 stages
# of S3D (Mixed_4b..5c: 60 % of the step's launches, 10-40 us kernels that the host cannot feed fast
# enough at the START of backward, when it has no lead) -- and leaves the large early stages eager.
_GRAPH_MODE = os.environ.get("COCLR_GRAPH_QUERY", "0")
GRAPH_QUERY = _GRAPH_MODE == "1"
GRAPH_LATE = _GRAPH_MODE == "late"
_GRAPH_WARMUP = 2           # eager passes with an unchanged signature before capturing
_STATIC_PTRS = set()        # addresses of static graph outputs (a later stage takes them in place)
_CAPTURE_STREAMS = {}


class _GraphEntry:
    __slots__ = ("sig", "seen", "fwd", "bwd", "pool", "x", "out", "dout", "grads", "dx", "run", "xin",
                 "version", "params", "need_dx", "disabled", "static_in", "static_dout")

    def __init__(self, sig):
        self.sig, self.seen = sig, 0
        self.fwd = self.bwd = self.pool = None
        self.version = 0
        self.disabled = False
        self.grads = self.dx = self.run = self.xin = self.x = self.out = self.dout = None
        self.static_in = self.static_dout = None
        self.params, self.need_dx = (), False


def _graph_signature(module, x, params, kwargs):
    bns = module.__dict__.get("_coclr_bn_list")
    if bns is None:
        bns = module.__dict__["_coclr_bn_list"] = [
            m for m in module.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    slots = tuple(_GRAD_SLOTS[id(p)][1].data_ptr() if id(p) in _GRAD_SLOTS else 0 for p in params)
    return (tuple(x.shape), x.dtype, x.device, bool(x.requires_grad), tuple(sorted(kwargs)),
            tuple(p.data_ptr() for p in params), tuple(bool(p.requires_grad) for p in params), slots,
            tuple((m.training, m.momentum, m.eps, m.running_mean.data_ptr()) for m in bns))


def _capture_stream(device):
    st = _CAPTURE_STREAMS.get(device)
    if st is None:
        st = _CAPTURE_STREAMS[device] = torch.cuda.Stream(device=device)
    return st


def _copy_rows(src, dst):
    """dst (dense) <- src (dense rows, possibly strided along dim 0) as one kernel."""
    ident = _IDENT.get((src.device, src.shape[0]))
    if ident is None:
        ident = _IDENT[(src.device, src.shape[0])] = torch.arange(src.shape[0], device=src.device)
    ops.gather_rows(src, ident, dst)


_IDENT = {}


def _graph_entry(module, x, params, kwargs):
    """The captured entry to replay for this call, or None (run eagerly)."""
    if kwargs or not x.is_cuda or x.dim() != 5 or torch.cuda.is_current_stream_capturing():
        return None
    store = module.__dict__.get("_coclr_graph_entries")
    if store is None:
        store = module.__dict__["_coclr_graph_entries"] = {}
    sig = _graph_signature(module, x, params, kwargs)
    key = (tuple(x.shape), bool(x.requires_grad))
    ent = store.get(key)
    if ent is None or ent.sig != sig:
        if ent is not None:                      # its static buffers go away with it
            for t in (ent.out, ent.dx):
                if t is not None:
                    _STATIC_PTRS.discard(t.data_ptr())
        ent = store[key] = _GraphEntry(sig)      # new shape / moved storage / flags changed: start over
    if ent.disabled:
        return None
    ent.seen += 1
    if ent.seen <= _GRAPH_WARMUP:
        return None
    if ent.fwd is None:
        try:
            _capture_forward(module, x, params, ent)
        except (RuntimeError, ValueError) as e:
            # e.g. an input whose rows are not dense: this (module, shape) stays on the eager path
            ent.disabled = True
            import warnings
            warnings.warn("coclr_amd: hipGraph capture of %s failed (%s); running it eagerly"
                          % (type(module).__name__, e))
            return None
    return ent


def _capture_forward(module, x, params, ent):
    dev = x.device
    need_dx = bool(x.requires_grad)
    cap = _capture_stream(dev)
    cur = torch.cuda.current_stream(dev)
    # the input of a later stage is the previous stage's static output: same address every step
    if _dense5(x) and x.data_ptr() in _STATIC_PTRS:
        static_x, ent.static_in = x.detach(), True
    else:
        static_x, ent.static_in = torch.empty(x.shape, dtype=x.dtype, device=dev), None
        _copy_rows(x.detach(), static_x)
    cur.synchronize()
    pool = torch.cuda.graph_pool_handle()
    g = torch.cuda.CUDAGraph()
    _STREAM_ALIAS[cap.cuda_stream] = cur.cuda_stream
    try:
        with torch.cuda.graph(g, pool=pool, stream=cap, capture_error_mode="thread_local"):
            run = Run(dev, save=True, need_input_grad=need_dx)
            xin = Val(static_x)
            if not need_dx:
                run.no_grad_bases.add(id(xin.base))
            run.begin(module)
            run.out = module._emit(run, xin)
            out = run.out.view()
    finally:
        _STREAM_ALIAS.pop(cap.cuda_stream, None)
    ent.fwd, ent.pool, ent.x, ent.out, ent.run, ent.xin = g, pool, static_x, out, run, xin
    ent.params, ent.need_dx = params, need_dx
    ent.bwd = None
    _STATIC_PTRS.add(out.data_ptr())


class GraphedFn(torch.autograd.Function):
    """EngineFn whose forward and backward are hipGraph replays (see GRAPH_QUERY)."""

    @staticmethod
    def forward(ctx, ent, x, *params):
        new_pass(x.device)
        if ent.static_in is None or x.data_ptr() != ent.x.data_ptr():
            _copy_rows(x.detach(), ent.x)
        ent.fwd.replay()
        ent.version += 1
        ctx.ent, ctx.version = ent, ent.version
        return ent.out.detach()

    @staticmethod
    def backward(ctx, dout):
        ent = ctx.ent
        if ctx.version != ent.version:
            raise RuntimeError(
                "coclr_amd: backward through a graph-replayed encoder pass whose activations a later "
                "forward has overwritten (two forwards, then two backwards); set COCLR_GRAPH_QUERY=0")
        ctx.ent = None
        params = ent.params
        # A caller that accumulates gradients still holds last step's `.grad`, which aliases the static
        # buffer this replay overwrites: give it its own memory first (rare: zero_grad() sets None)
        if ent.grads is not None:
            for p, g in zip(params, ent.grads):
                if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                    p.grad = p.grad.clone()
        if ent.bwd is None:
            _capture_backward(ent, dout)
        elif ent.static_dout is None or dout.data_ptr() != ent.dout.data_ptr():
            ent.dout.copy_(dout)
        ent.bwd.replay()
        grads = tuple(None if g is None else g.view_as(g) for g in ent.grads)
        dx = ent.dx.view_as(ent.dx) if ent.dx is not None else None
        return (None, dx) + grads


def _capture_backward(ent, dout):
    dev = dout.device
    cap = _capture_stream(dev)
    cur = torch.cuda.current_stream(dev)
    # the gradient of an earlier stage's output is a later stage's static dx: same address every step
    if dout.is_contiguous() and dout.data_ptr() in _STATIC_PTRS:
        static_dout, ent.static_dout = dout.detach(), True
    else:
        static_dout, ent.static_dout = torch.empty(ent.out.shape, dtype=dout.dtype, device=dev), None
        static_dout.copy_(dout)
    cur.synchronize()
    run = ent.run
    g = torch.cuda.CUDAGraph()
    _STREAM_ALIAS[cap.cuda_stream] = cur.cuda_stream
    try:
        with torch.cuda.graph(g, pool=ent.pool, stream=cap, capture_error_mode="thread_local"):
            run.backward(static_dout)
            dx = run.grads.pop(id(ent.xin.base), None) if ent.need_dx else None
            grads = tuple(run.param_grads.pop(id(p), None) if p.requires_grad else None
                          for p in ent.params)
            run.param_grads.clear()
            run.grads.clear()
    finally:
        _STREAM_ALIAS.pop(cap.cuda_stream, None)
    ent.bwd, ent.dout, ent.grads, ent.dx = g, static_dout, grads, dx
    ent.run = None           # the tape has been consumed; its tensors live on in the graphs' pool
    if dx is not None:
        _STATIC_PTRS.add(dx.data_ptr())


# ---------------------------------------------------------------------------------
# Launch-plan replay of a DIFFERENTIATED pass (coclr_amd/plan.py)
# ---------------------------------------------------------------------------------
# Same idea as the hipGraph replay above -- the launch sequence of a node is static per signature -- without
# its device-side cost: the recorded C-ABI calls are re-issued as ordinary launches on the ordinary streams.
# After _PLAN_WARMUP interpreted passes with an unchanged signature the node's forward is run once more with
# its allocations in a private torch.cuda.MemPool while every call it makes is logged; its tape is kept (the
# tensors it references now have addresses nobody else is given) and logged the same way at the first backward.
# From then on a step of the node is: patch the two places that hold the input's address if the caller's
# tensor moved, re-issue the forward log; copy dout into its recorded place, re-issue the backward log, hand
# autograd fresh aliases of the recorded gradient tensors (DDP bucket views where DDP has them).
# COCLR_PLAN=0 switches it off (the interpreted pass); hipGraph replay (COCLR_GRAPH_QUERY) takes precedence.
PLAN = os.environ.get("COCLR_PLAN", "1") != "0" and hasattr(torch.cuda, "MemPool") and \
    hasattr(torch.cuda, "use_mem_pool")        # (a torch without private pools: every pass stays interpreted)
_PLAN_WARMUP = 4            # interpreted passes first: one-time allocations, DDP's bucket rebuild, slot verification
_PLAN_MAX_SHAPES = 3        # input shapes per node that get a plan (and a pool of activations) of their own
PLAN_STATS = {"recorded": 0, "replayed": 0, "disabled": []}


class _PlanEntry:
    __slots__ = ("sig", "seen", "pool", "fwd", "bwd", "run", "xin", "x_ptr", "x_span", "x_refs", "x_fixed",
                 "out", "dout", "grads", "dx", "params", "need_dx", "version", "disabled", "static_dout", "defer",
                 "stream")

    def __init__(self, sig):
        self.sig, self.seen = sig, 0
        self.pool = self.fwd = self.bwd = self.run = self.xin = None
        self.x_ptr = self.x_span = 0
        self.x_refs, self.x_fixed = None, False
        self.out = self.dout = self.grads = self.dx = None
        self.params, self.need_dx = (), False
        self.version = 0
        self.disabled = False
        self.static_dout = None
        self.defer = False
        self.stream = 0


def _plan_signature(module, x, params):
    return _graph_signature(module, x, params, {}) + (
        tuple(x.stride()), WGRAD_STREAM, DEFER_JOIN, PAIR_UNITS, FUSE_BN_REDUCE, LAZY_APPLY, POOLED_BACKWARD,
        BATCH_PACK, tuple(id(p) in _SLOTS_VERIFIED for p in params))


def _plan_drop(ent):
    for t in (ent.out, ent.dx):
        if t is not None:
            _STATIC_PTRS.discard(t.data_ptr())


def _plan_entry(module, x, params, kwargs):
    """The node's plan entry when this call can go through it (recording or replaying), else None."""
    if kwargs or not x.is_cuda or x.dim() != 5 or not _dense5(x) or LANES or DECISION_PROBE is not None or \
            torch.cuda.is_current_stream_capturing():
        return None
    store = module.__dict__.get("_coclr_plan_entries")
    if store is None:
        store = module.__dict__["_coclr_plan_entries"] = {}
    sig = _plan_signature(module, x, params)
    key = (tuple(x.shape), bool(x.requires_grad))
    ent = store.get(key)
    if ent is None and len(store) >= _PLAN_MAX_SHAPES:
        return None                              # every entry keeps a memory pool: odd shapes run interpreted
    if ent is None or ent.sig != sig:
        if ent is not None:
            _plan_drop(ent)
        ent = store[key] = _PlanEntry(sig)       # new shape / moved storage / switches changed: start over
    if ent.disabled:
        return None
    ent.seen += 1
    if ent.seen <= _PLAN_WARMUP:
        return None
    stream = torch.cuda.current_stream(x.device).cuda_stream
    if ent.fwd is not None:
        if stream != ent.stream:
            return None                          # recorded on another stream: this call runs interpreted
        if x.data_ptr() != ent.x_ptr and (ent.x_fixed or ent.run is not None):
            # the input moved and the logs cannot follow it: an address inside a multi-problem table, or a tape
            # that has not been logged yet (forward recorded, no backward since) and reads the input through
            # its own tensor objects.  This call runs interpreted.
            return None
    return ent


def _record_forward(ent, module, x, defer):
    dev = x.device
    need_dx = bool(x.requires_grad)
    ent.pool = torch.cuda.MemPool()
    ent.stream = torch.cuda.current_stream(dev).cuda_stream
    rec = _plan.Recorder(ent.stream)
    with torch.cuda.use_mem_pool(ent.pool, device=dev), rec:
        run = Run(dev, save=True, need_input_grad=need_dx)
        xin = Val(x.detach())
        if not need_dx:
            run.no_grad_bases.add(id(xin.base))
        run.begin(module)
        run.out = module._emit(run, xin)
        out = run.out.view()
        if not out.is_contiguous():
            rec.taint("the node's output is a channel slice")
            out = out.contiguous()
    n, c, t, h, w = x.shape
    ent.x_ptr = x.data_ptr()
    ent.x_span = ((n - 1) * x.stride(0) + c * t * h * w) * x.element_size()
    ent.fwd, ent.run, ent.xin, ent.out = rec.plan, run, xin, out
    ent.params, ent.need_dx, ent.defer = None, need_dx, defer
    ent.x_refs = [rec.plan.pointer_refs(ent.x_ptr, ent.x_ptr + ent.x_span), None]
    ent.x_fixed = rec.plan.embedded_refs(ent.x_ptr, ent.x_ptr + ent.x_span)
    if rec.tainted:
        ent.disabled = True
        PLAN_STATS["disabled"].append(rec.tainted)
    PLAN_STATS["recorded"] += 1
    _STATIC_PTRS.add(out.data_ptr())


def _record_backward(ent, dout, params):
    dev = dout.device
    if dout.is_contiguous() and dout.data_ptr() in _STATIC_PTRS:
        static_dout, ent.static_dout = dout.detach(), True
    else:
        static_dout, ent.static_dout = None, None
    run = ent.run
    rec = _plan.Recorder(torch.cuda.current_stream(dev).cuda_stream)
    with torch.cuda.use_mem_pool(ent.pool, device=dev):
        if static_dout is None:
            static_dout = torch.empty(ent.out.shape, dtype=dout.dtype, device=dev)
            static_dout.copy_(dout)
        with rec:
            run.backward(static_dout, defer_join=DEFER_JOIN and ent.defer)
            dx = run.grads.pop(id(ent.xin.base), None) if ent.need_dx else None
            grads = tuple(run.param_grads.pop(id(p), None) if p.requires_grad else None for p in params)
            run.param_grads.clear()
            run.grads.clear()
    if rec.plan.stream != ent.stream:
        rec.taint("backward ran on another stream than the forward")
    ent.bwd, ent.dout, ent.grads, ent.dx = rec.plan, static_dout, grads, dx
    ent.run = None           # the tape has been consumed; its tensors live on in the entry's pool
    ent.x_refs[1] = rec.plan.pointer_refs(ent.x_ptr, ent.x_ptr + ent.x_span)
    ent.x_fixed = ent.x_fixed or rec.plan.embedded_refs(ent.x_ptr, ent.x_ptr + ent.x_span)
    if rec.tainted:
        ent.disabled = True
        PLAN_STATS["disabled"].append(rec.tainted)
    if dx is not None:
        _STATIC_PTRS.add(dx.data_ptr())


class PlanFn(torch.autograd.Function):
    """EngineFn whose forward and backward re-issue recorded launch plans (see PLAN)."""

    @staticmethod
    def forward(ctx, ent, module, x, *params):
        new_pass(x.device)
        if ent.fwd is None:
            _record_forward(ent, module, x, getattr(module, "__dict__", {}).get("_coclr_defer_join", False))
        else:
            p = x.data_ptr()
            if p != ent.x_ptr:
                ent.fwd.patch(ent.x_refs[0], p)
                if ent.bwd is not None:
                    ent.bwd.patch(ent.x_refs[1], p)
                ent.x_ptr = p
            ent.fwd.replay()
            PLAN_STATS["replayed"] += 1
        ent.version += 1
        ctx.ent, ctx.version, ctx.params = ent, ent.version, params
        return ent.out.detach()

    @staticmethod
    def backward(ctx, dout):
        ent = ctx.ent
        if ent is None:
            raise RuntimeError("coclr_amd: backbone backward called twice (graph not retained)")
        if ctx.version != ent.version:
            raise RuntimeError(
                "coclr_amd: backward through a planned encoder pass whose activations a later forward has "
                "overwritten (two forwards, then two backwards); set COCLR_PLAN=0")
        ctx.ent = None
        params = ctx.params
        # a caller that accumulates gradients still holds last step's `.grad`, which aliases the recorded
        # tensor this pass overwrites: give it its own memory first (rare: zero_grad() sets None)
        if ent.grads is not None:
            for p, g in zip(params, ent.grads):
                if g is not None and p.grad is not None and p.grad.data_ptr() == g.data_ptr():
                    p.grad = p.grad.clone()
        if ent.bwd is None:
            _record_backward(ent, dout, params)
        else:
            if torch.cuda.current_stream(dout.device).cuda_stream != ent.stream:
                raise RuntimeError("coclr_amd: planned backward called on another stream than it was recorded on")
            if ent.static_dout is None or dout.data_ptr() != ent.dout.data_ptr():
                ent.dout.copy_(dout)
            ent.bwd.replay()
        grads = tuple(None if g is None else g.view_as(g) for g in ent.grads)
        dx = ent.dx.view_as(ent.dx) if ent.dx is not None else None
        return (None, None, dx) + grads


def _dense5(t):
    n, c, d, h, w = t.shape
    s = t.stride()
    return s[4] == 1 and s[3] == w and s[2] == h * w and s[1] == d * h * w


def run_module(module, x, **kwargs):
    """Forward `module` (anything with `_emit(run, val, **kw)`) on x."""
    params = module.__dict__.get("_coclr_params")
    if params is None:
        params = module.__dict__["_coclr_params"] = list(module.parameters())
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        if GRAPH_QUERY or (GRAPH_LATE and module.__dict__.get("_coclr_graph_late")):
            ent = _graph_entry(module, x, params, kwargs)
            if ent is not None:
                return GraphedFn.apply(ent, x, *params)
        elif PLAN:
            ent = _plan_entry(module, x, params, kwargs)
            if ent is not None:
                return PlanFn.apply(ent, module, x, *params)
        return EngineFn.apply(module, kwargs, x, *params)
    run = Run(x.device, save=False)
    xin = Val(x if _dense5(x) else x.contiguous())
    run.begin(module)
    out = module._emit(run, xin, **kwargs).view()
    plan = run.plan
    if plan is not None and plan.dirty and not (
            x.device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        plan.build(x.device)      # ready before a caller captures the next pass into a hipGraph
    return out if out.is_contiguous() else out.contiguous()
