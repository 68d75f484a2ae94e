"""What makes bench.py's multi-GPU run un-failable and self-verifying.

Three layers, outermost first:

1. PROCESS LADDER (`supervise`).  At world > 1 every rank that the driver launches is a SUPERVISOR: it
   never touches a GPU, it starts the measuring process (bench.py again, COCLR_BENCH_CHILD=1) as a child,
   and the supervisors agree over the launcher's TCP store on what happened.  A child that hangs (its
   watchdog names the exchange and ends it), dies (fault in a kernel that reads a peer's memory, RCCL
   error) or raises takes its HIP context and its RCCL communicators with it -- the only re-initialisation
   that can be trusted after a GPU-side failure -- and every supervisor starts the next attempt one or more
   rungs further down the ladder.  Rank 0's supervisor prints the contract's single JSON line: the
   child's, with `multi_gpu.attempts` saying what was tried.

2. SELF-CHECK (`SelfCheck`).  Replicas that agree with each other prove nothing about ORDER: a gradient
   all-reduced before its last weight-gradient kernel has landed is wrong identically on every rank.  So
   inside the child, after enough steps for the fast path to be in its steady state (bucket views
   published and verified, joins deferred, graphs captured, exchange scheme chosen): snapshot everything
   a step changes, run ONE step in the configuration to be timed, restore, run the same step in the SERIAL
   configuration -- no deferred join, DDP's own gradient copies (a second wrapper without the bucket hook,
   its buckets rebuilt like the first one's), the reference's all-gather exchange, no graph replay, every
   kernel on ONE stream -- and require parameters, queues, BatchNorm buffers and Adam moments after the
   two steps to be BIT-identical on every rank.  Same kernels, same operands, same bucket layout: any
   difference is an ordering bug (or a nondeterministic kernel), and it cannot hide.

3. IN-PROCESS RUNGS.  On a mismatch the child does not give up the attempt: it walks down the rungs in
   process (each is a run-time switch), re-running the checked step until one matches the serial
   reference, and times THAT configuration; `multi_gpu.rung` names what had to be switched off.

Rungs (cumulative):
    0 fast      everything on (exchange scheme "auto": peer row pull if verified, else routed all-to-all)
    1 -defer    every stage node joins the weight-gradient stream (COCLR_DEFER_JOIN=0)
    2 -hook     DDP's own per-parameter bucket copies, no engine writes into bucket views (COCLR_DDP_HOOK=0)
    3 -pull     shuffle-BN exchange = RCCL all_to_all_single (COCLR_SHUFFLE=routed)
    4 -routed   shuffle-BN exchange = the reference's all-gather (COCLR_SHUFFLE=allgather)
    5 -graphs   key encoder launched eagerly (COCLR_GRAPHS=0), query encoder interpreted (COCLR_PLAN=0)
    6 serial    one stream: no key-encoder stream, no weight-gradient stream; the queues re-sent from rank 0 with
                every forward, as DDP(broadcast_buffers=True) does
"""
import json
import os
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

RUNG_NAMES = ["fast", "-defer", "-hook", "-pull", "-routed", "-graphs", "serial"]
SERIAL = len(RUNG_NAMES) - 1
RUNG_WHAT = {
    0: "everything on",
    1: "deferred joins of the weight-gradient stream off",
    2: "+ bucket hook off (DDP's own per-parameter copies)",
    3: "+ peer row pull off (routed all_to_all_single)",
    4: "+ routed exchange off (the reference's all-gather)",
    5: "+ hipGraph replay of the key encoder off",
    6: "+ side streams off (every kernel on one stream), queues in every buffer broadcast",
}
FLOOR_HANG = False       # fault injection (tests): see inject_fault("floor_hangs")
EXIT_MISMATCH = 7        # the self-check failed on the last rung as well
EXIT_WATCHDOG = 5


# ---------------------------------------------------------------------------------------------------
# run-time configuration switches
# ---------------------------------------------------------------------------------------------------

class Switches:
    """The product's run-time switches, as they were when the process started (environment) and as a rung
    leaves them.  A rung only ever turns things OFF relative to the starting configuration."""

    def __init__(self):
        from coclr_amd import engine
        import coclr_amd.model.pretrain as impl
        self.engine, self.impl = engine, impl
        self.base = {"defer": engine.DEFER_JOIN, "shuffle": impl._SHUFFLE_MODE, "graphs": impl._GRAPHS,
                     "wgrad_stream": engine.WGRAD_STREAM, "overlap_keys": impl._OVERLAP_KEYS,
                     "plan": engine.PLAN,
                     "sync_queues": impl._SYNC_QUEUES,
                     "hook": os.environ.get("COCLR_DDP_HOOK", "1") != "0"
                     and os.environ.get("COCLR_PATCH_DDP", "1") != "0"}
        self.rung = 0
        self.decided = None      # what COCLR_SHUFFLE=auto turned into (read off the product while no rung forces it)

    def wants_hook(self, rung):
        return self.base["hook"] and rung < 2

    def apply(self, rung):
        e, m, b = self.engine, self.impl, self.base
        e.DEFER_JOIN = b["defer"] and rung < 1
        if self.rung < 3 and m._SHUFFLE_MODE != "auto":
            self.decided = m._SHUFFLE_MODE               # the product's own decision (or its fallback)
        shuffle = self.decided or b["shuffle"]
        if rung >= 3 and shuffle in ("auto", "pull"):
            shuffle = "routed"
        if rung >= 4:
            shuffle = "allgather"
        m._SHUFFLE_MODE = shuffle
        m._GRAPHS = b["graphs"] and rung < 5
        e.PLAN = b["plan"] and rung < 5          # launch-plan replay of the query encoder goes with the graphs
        e.WGRAD_STREAM = b["wgrad_stream"] and rung < 6
        m._OVERLAP_KEYS = b["overlap_keys"] and rung < 6
        m._SYNC_QUEUES = b["sync_queues"] or rung >= 6      # serial: queues re-sent with every forward (the reference)
        self.rung = rung

    def describe(self):
        e, m = self.engine, self.impl
        return {"defer_join": bool(e.DEFER_JOIN), "shuffle": m._SHUFFLE_MODE, "graphs": bool(m._GRAPHS),
                "launch_plans": bool(e.PLAN), "wgrad_stream": bool(e.WGRAD_STREAM), "key_stream": bool(m._OVERLAP_KEYS),
                "queues_in_every_broadcast": bool(m._SYNC_QUEUES)}


def rung_env(rung):
    """Environment of a child that STARTS on `rung` (the child applies the rung at run time as well; the
    environment makes the import-time defaults agree, e.g. no hook is ever registered from rung 2 on)."""
    env = {"COCLR_BENCH_RUNG": str(rung)}
    if rung >= 2:
        env["COCLR_DDP_HOOK"] = "0"
    return env


# ---------------------------------------------------------------------------------------------------
# snapshot / restore / compare
# ---------------------------------------------------------------------------------------------------

def state_tensors(model, opt):
    """Every tensor a training step changes: parameters, buffers (queues, pointer, BatchNorm statistics and
    counters) and the optimiser's moments and step counters -- (name, tensor), each storage once."""
    out, seen = [], set()
    for k, v in model.state_dict().items():
        key = (v.data_ptr(), v.numel(), v.dtype)
        if v.numel() and key in seen:
            continue                                      # S3D's alias keys name the same tensor twice
        seen.add(key)
        out.append(("model." + k, v))
    for gi, g in enumerate(opt.param_groups):
        for pi, p in enumerate(g["params"]):
            st = opt.state.get(p)
            if st:
                for key in sorted(st):
                    if torch.is_tensor(st[key]):
                        out.append(("adam.%d.%d.%s" % (gi, pi, key), st[key]))
    return out


class Snapshot:
    def __init__(self, model, opt):
        self.items = [(n, t, t.detach().clone()) for n, t in state_tensors(model, opt)]
        self.rng = torch.get_rng_state()
        self.folded = model.__dict__.get("_momentum_folded")
        self.model = model

    @torch.no_grad()
    def restore(self):
        for _, live, saved in self.items:
            live.copy_(saved)
        torch.set_rng_state(self.rng)
        if self.folded is None:
            self.model.__dict__.pop("_momentum_folded", None)
        else:
            self.model.__dict__["_momentum_folded"] = self.folded


def capture(model, opt):
    return [(n, t.detach().clone()) for n, t in state_tensors(model, opt)]


def compare(a, b):
    """Bit-for-bit comparison of two captures.  NaN never equals NaN under torch.equal, so a non-finite
    state fails here too; it is reported separately."""
    bad, first, worst, finite = 0, None, 0.0, True
    if len(a) != len(b):
        return {"identical": False, "tensors": len(a), "differing": abs(len(a) - len(b)),
                "first_mismatch": "different state layouts", "max_abs_diff": None, "finite": False}
    for (n, x), (_, y) in zip(a, b):
        if x.is_floating_point() and not bool(torch.isfinite(x).all() and torch.isfinite(y).all()):
            finite = False
        if x.shape != y.shape or not torch.equal(x, y):
            bad += 1
            if first is None:
                first = n
            if x.shape == y.shape and x.numel():
                d = float((x.double() - y.double()).abs().max())
                worst = max(worst, d) if d == d else float("inf")
    return {"identical": bad == 0 and finite, "tensors": len(a), "differing": bad, "first_mismatch": first,
            "max_abs_diff": worst, "finite": finite}


def agree(ok, device):
    """MIN over ranks of a local verdict."""
    t = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() == 1.0)


# ---------------------------------------------------------------------------------------------------
# the self-check
# ---------------------------------------------------------------------------------------------------

class SelfCheck:
    """Drives the checked step of layer 2 / 3 above.  `make_ddp(hook)` returns the wrapper to run a rung
    on (built once each); `run_step(ddp, i)` runs one full training step on batch i."""

    def __init__(self, model, opt, make_ddp, run_step, device, switches, sync):
        self.model, self.opt, self.make_ddp, self.run_step = model, opt, make_ddp, run_step
        self.device, self.sw, self.sync = device, switches, sync
        self._plain_ready = False

    def _wrapper(self, rung):
        from coclr_amd import engine
        hook = self.sw.wants_hook(rung)
        if not hook:
            # no engine writes into the hooked wrapper's bucket views while the plain wrapper runs
            engine._GRAD_SLOTS.clear()
            engine._SLOTS_VERIFIED.clear()
        return self.make_ddp(hook)

    def _step_on(self, rung, snap, batch):
        """restore -> one step on `rung` -> capture"""
        snap.restore()
        self.sw.apply(rung)
        self.run_step(self._wrapper(rung), batch)
        self.sync()
        return capture(self.model, self.opt)

    def run(self, rung, batch=0, dog=None):
        from coclr_amd import engine
        sw = self.sw
        snap = Snapshot(self.model, self.opt)
        deferred0 = engine.DEFERRED[0]
        if dog is not None:
            dog.at("self-check: the step in the configuration to be timed (rung %d)" % rung)
        fast = self._step_on(rung, snap, batch)
        deferred = engine.DEFERRED[0] - deferred0
        # the serial reference: DDP rebuilds its buckets (gradient-ready order) after a wrapper's first
        # backward pass -- two throw-away steps put the plain wrapper on the layout the hooked one has
        if dog is not None:
            dog.at("self-check: the serial reference step")
        if not self._plain_ready:
            snap.restore()
            sw.apply(SERIAL)
            plain = self._wrapper(SERIAL)
            for i in range(2):
                self.run_step(plain, batch + 1 + i)
            self.sync()
            self._plain_ready = True
        ref = self._step_on(SERIAL, snap, batch)
        trials = []
        chosen = None
        r = rung
        cur = fast
        while True:
            c = compare(cur, ref)
            ok = agree(c["identical"], self.device)
            trials.append({"rung": r, "name": RUNG_NAMES[r], "bit_identical_to_serial_on_every_rank": ok,
                           "this_rank": {k: c[k] for k in ("differing", "first_mismatch", "max_abs_diff",
                                                           "finite")}})
            if ok:
                chosen = r
                break
            if r >= SERIAL:
                break
            r += 1
            if dog is not None:
                dog.at("self-check: mismatch, retrying the step on rung %d (%s)" % (r, RUNG_NAMES[r]))
            cur = self._step_on(r, snap, batch)
        final = chosen if chosen is not None else SERIAL
        snap.restore()
        sw.apply(final)
        return {"passed": chosen is not None, "rung": final, "rung_name": RUNG_NAMES[final],
                "started_on_rung": rung, "trials": trials, "tensors_compared": len(ref),
                "deferred_nodes_in_checked_step": deferred, "switches": sw.describe(),
                "what": "one steady-state training step run twice from the same snapshot (parameters, "
                        "buffers, Adam state, host RNG): in the configuration to be timed and in the SERIAL "
                        "one (joins not deferred, DDP's own bucket copies through a second wrapper, all-gather "
                        "exchange, no graph replay, one stream); everything the step changes must be "
                        "bit-identical on every rank, else the next rung is tried in process"}, final


# ---------------------------------------------------------------------------------------------------
# fault injection (tests only: COCLR_BENCH_FAULT)
# ---------------------------------------------------------------------------------------------------

def inject_fault(kind, rank):
    """Faults for the ladder's tests, each tied to the switch whose failure it stands for -- so that the
    bench is expected to END on the rung that removes it.  Never set outside tests/."""
    from coclr_amd import engine, parallel
    import coclr_amd.model.pretrain as impl
    if kind == "defer":
        # a deferred join that loses a gradient: only reachable while joins are deferred.  (On the CPU
        # double nothing defers -- no streams -- so the fault keys on the switch itself.)
        orig = engine.Run.backward

        def backward(self, dout, defer_join=False):
            out = orig(self, dout, defer_join=defer_join)
            if engine.DEFER_JOIN and defer_join and self.param_grads:
                k = next(iter(self.param_grads))
                self.param_grads[k].add_(1.0)
            return out
        engine.Run.backward = backward
    elif kind == "hook":
        # the bucket hook all-reduces a bucket one of whose gradients is not there yet
        orig_hook = parallel.bucket_hook

        def hook(state, bucket):
            if bucket.index() == 0:
                bucket.buffer()[:1].add_(1.0)
            return orig_hook(state, bucket)
        parallel.bucket_hook = hook
    elif kind == "routed_raises":
        def a2a(*a, **k):
            raise RuntimeError("all_to_all_single: injected backend failure")
        dist.all_to_all_single = a2a
    elif kind == "routed_hangs":
        native = dist.all_to_all_single

        def a2a(*a, **k):
            if rank == 1:
                time.sleep(3600)
            return native(*a, **k)
        dist.all_to_all_single = a2a
    elif kind == "graphs":
        orig_enc = impl.InfoNCE._encode_graphed

        def enc(self, encoder, src, n_index, pre=None):
            out = orig_enc(self, encoder, src, n_index, pre=pre)
            return out + 1e-3 if impl._GRAPHS else out
        impl.InfoNCE._encode_graphed = enc
    elif kind == "pull_map":
        import torch.multiprocessing.reductions as red

        def refuse(t):
            raise RuntimeError("hipIpcGetMemHandle: invalid argument (injected)")
        if rank == 1:
            red.reduce_tensor = refuse
    elif kind == "floor_hangs":
        # one rank never comes back from the OPTIONAL host-floor leg that follows the timed region
        global FLOOR_HANG
        FLOOR_HANG = rank == 1
    elif kind == "dies":
        # a rank that disappears in its first exchange on the fast rungs (stands for a GPU memory fault)
        native = dist.all_to_all_single

        def a2a(*a, **k):
            if rank == 1:
                os._exit(134)
            return native(*a, **k)
        dist.all_to_all_single = a2a
    elif kind:
        raise SystemExit("unknown COCLR_BENCH_FAULT=%r" % kind)


# ---------------------------------------------------------------------------------------------------
# the process ladder
# ---------------------------------------------------------------------------------------------------

def _store():
    """Client of the launcher's TCP store (torchrun's agent hosts it on MASTER_PORT); rank 0 hosts it when
    the ranks were started by hand."""
    import datetime
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29577"))
    world = int(os.environ["WORLD_SIZE"])
    rank = int(os.environ["RANK"])
    agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"
    return dist.TCPStore(host, port, world, is_master=(rank == 0 and not agent),
                         timeout=datetime.timedelta(seconds=600), wait_for_workers=False)


def _wait_keys(store, keys, seconds):
    """store.wait without its fixed timeout: poll until every key is there (True) or `seconds` have passed."""
    end = time.monotonic() + seconds
    while time.monotonic() < end:
        if store.check(list(keys)):
            return True
        time.sleep(0.25)
    return False


def next_rung(rung, statuses):
    """Where the next attempt starts, from what the failed one left behind.  A failure that names the
    exchange it happened in drops to the rung that removes that exchange; an unattributed one first gives up
    the one path that has never run on this hardware (the peer pull), then everything."""
    names = " | ".join((s.get("last_collective") or "") + " " + (s.get("error") or "") for s in statuses)
    if "_pull_shuffle" in names or "_peer_stage" in names or "_auto_shuffle" in names or "hipIpc" in names:
        want = 3
    elif "all_to_all" in names:
        want = 4
    elif "ddp bucket" in names:
        want = 2
    elif rung < 3:
        want = 3
    else:
        want = SERIAL
    return max(want, rung + 1) if rung < SERIAL else None


def supervise(argv, hang_timeout, store=None):
    """Layer 1.  Returns the process exit code."""
    import signal
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        signal.signal(sig, _on_term)
    if store is None:
        store = _store()
    pre = "coclr_bench/%s/" % os.environ.get("TORCHELASTIC_RUN_ID", "run")
    rung = int(os.environ.get("COCLR_BENCH_RUNG", "0"))
    attempts = []
    attempt_dirs = []
    final_line = None
    succeeded = False
    max_attempts = int(os.environ.get("COCLR_BENCH_ATTEMPTS", "3"))
    for attempt in range(max_attempts):
        # ---- a directory for this attempt (rendezvous file of the children, their status files) -------
        if rank == 0:
            adir = tempfile.mkdtemp(prefix="coclr_bench_a%d_" % attempt)
            attempt_dirs.append(adir)
            store.set(pre + "%d/dir" % attempt, adir)
        else:
            if not _wait_keys(store, [pre + "%d/dir" % attempt], 1800):
                raise RuntimeError("bench supervisor: rank 0's supervisor never opened attempt %d" % attempt)
            adir = store.get(pre + "%d/dir" % attempt).decode()
        env = dict(os.environ, COCLR_BENCH_CHILD="1", COCLR_BENCH_DIR=adir,
                   COCLR_BENCH_INIT="file://" + os.path.join(adir, "rendezvous"), **rung_env(rung))
        child = subprocess.Popen([sys.executable, os.path.abspath(sys.argv[0])] + argv, env=env,
                                 stdout=subprocess.PIPE, stderr=None, text=True, start_new_session=True)
        _CHILDREN.append(child)
        # the child's own watchdog ends a hang after `hang_timeout` without progress; this limit is the
        # backstop for a child that cannot even do that
        limit = time.monotonic() + float(os.environ.get("COCLR_BENCH_ATTEMPT_LIMIT", 4 * hang_timeout + 600))
        out, peer_failed_at, killed = None, None, False
        keys = [pre + "%d/status/%d" % (attempt, r) for r in range(world)]
        while True:
            try:
                out, _ = child.communicate(timeout=0.5)
                break
            except subprocess.TimeoutExpired:
                pass
            now = time.monotonic()
            # a peer's child has failed: mine is waiting for a rank that no longer exists
            if peer_failed_at is None:
                for r in range(world):
                    if r != rank and store.check([keys[r]]):
                        st = json.loads(store.get(keys[r]).decode())
                        if st["rc"] != 0:
                            peer_failed_at = now
                            break
            if now > limit or (peer_failed_at is not None and now - peer_failed_at > 15.0):
                _kill(child)
                killed = True
                out, _ = child.communicate()
                break
        rc = child.returncode
        status = {"rank": rank, "rc": rc}
        sfile = os.path.join(adir, "status.%d.json" % rank)
        if os.path.exists(sfile):
            try:
                status.update(json.load(open(sfile)))
            except Exception:
                pass
        if killed:
            status["killed_after_peer_failure" if peer_failed_at is not None else "killed_at_attempt_limit"] = True
        line = None
        for ln in (out or "").splitlines():
            if ln.startswith("{"):
                line = ln
            elif ln.strip():
                print(ln, file=sys.stderr)             # RCCL's banner etc.: never on the supervisor's stdout
        store.set(keys[rank], json.dumps(status))
        # every supervisor reports within the attempt limit (a child cannot outlive it); peers that do not are
        # counted as failed rather than waited for forever
        _wait_keys(store, keys, float(os.environ.get("COCLR_BENCH_ATTEMPT_LIMIT", 4 * hang_timeout + 600)) + 60)
        statuses = [json.loads(store.get(k).decode()) if store.check([k]) else
                    {"rank": r, "rc": -1, "error": "its supervisor did not report"} for r, k in enumerate(keys)]
        ok = all(s["rc"] == 0 for s in statuses)
        rec = None
        if rank == 0:
            try:
                rec = json.loads(line) if line else None
            except Exception:
                rec = None
            ok = ok and rec is not None and rec.get("value") is not None
            store.set(pre + "%d/verdict" % attempt, "1" if ok else "0")
        else:
            ok = _wait_keys(store, [pre + "%d/verdict" % attempt], 600) and \
                store.get(pre + "%d/verdict" % attempt) == b"1"
        summary = {"attempt": attempt, "started_on_rung": rung, "rung_name": RUNG_NAMES[rung], "ok": ok,
                   "exit_codes": [s["rc"] for s in statuses]}
        if not ok:
            culprits = [s for s in statuses if s["rc"] != 0 and not s.get("killed_after_peer_failure")]
            summary["failed_ranks"] = [{k: s.get(k) for k in ("rank", "rc", "phase", "last_collective", "error")}
                                       for s in (culprits or statuses)[:4]]
        attempts.append(summary)
        if ok:
            succeeded = True
            if rank == 0:
                rec.setdefault("multi_gpu", {})["attempts"] = attempts
                final_line = json.dumps(rec)
            break
        if rank == 0:
            print("bench supervisor: attempt %d on rung %d (%s) failed: %s"
                  % (attempt, rung, RUNG_NAMES[rung], json.dumps(summary.get("failed_ranks"))),
                  file=sys.stderr, flush=True)
        nxt = next_rung(rung, [s for s in statuses if s["rc"] != 0])
        if nxt is None:
            break
        if attempt + 2 == max_attempts and attempt >= 1:
            nxt = SERIAL             # the last attempt allowed: everything off
        rung = nxt
    if rank == 0:
        if final_line is None:
            base = {"metric": "clips/sec (whole node)", "value": None, "unit": "clips/sec", "n_gpus": world,
                    "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "fp32", "data": "synthetic",
                    "multi_gpu": {"attempts": attempts, "error": "every rung of the ladder failed"}}
            final_line = json.dumps(base)
        sys.stdout.flush()
        print(final_line, flush=True)
    # leave together: the store lives in the launcher (or in rank 0, which therefore leaves last)
    store.set(pre + "done/%d" % rank, "1")
    if rank == 0:
        _wait_keys(store, [pre + "done/%d" % r for r in range(world)], 120)
        import shutil
        for d in attempt_dirs:               # rendezvous files and status files of the attempts
            shutil.rmtree(d, ignore_errors=True)
    return 0 if succeeded else 1


_CHILDREN = []


def _on_term(signum, frame):
    for c in _CHILDREN:
        if c.poll() is None:
            _kill(c)
    os._exit(128 + signum)


def _kill(child):
    """End exactly the process group this supervisor started (start_new_session: the child leads its own
    session, so exactly it and what it started can be ended).  The other direction -- a supervisor that is
    killed must not leave a rank on a GPU -- is the child's job: its watchdog thread ends the process when its
    parent changes (bench.Watchdog), and SIGTERM to a supervisor is forwarded (_on_term).  (No preexec_fn:
    running Python between fork and exec in a process that has torch's threads can deadlock.)"""
    import signal
    try:
        os.killpg(child.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        try:
            child.kill()
        except Exception:
            pass


def write_status(**kw):
    """Child side: what the supervisor reads if this process does not end by itself."""
    d = os.environ.get("COCLR_BENCH_DIR")
    if not d:
        return
    path = os.path.join(d, "status.%s.json" % os.environ.get("RANK", "0"))
    try:
        with open(path + ".tmp", "w") as f:
            json.dump(kw, f)
        os.replace(path + ".tmp", path)
    except OSError:
        pass
