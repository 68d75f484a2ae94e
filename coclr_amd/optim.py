"""torch.optim.Adam for the launch scripts' parameter structure, as one HIP launch.

main_nce.py:190-200 / main_coclr.py:205-213 build ONE param group per tensor (470 groups for
S3D InfoNCE, 235 of them with gradients) and call `optimizer.step()` (main_nce.py:331): torch's
implementation then runs its per-group loop 470 times -- hundreds of small launches and ~10 ms
of host time per step.  `Adam` below is a drop-in subclass (same constructor, same
`state_dict()` format: per-parameter `step` / `exp_avg` / `exp_avg_sq`) whose `step()` is a
single pointer-table kernel over every tensor (`coclr_adam_step`, csrc/optim.hip).

`install()` (called by the `model.pretrain` shim unless COCLR_PATCH_ADAM=0) makes
`torch.optim.Adam` resolve to this class, so the unmodified launch scripts pick it up.  Anything the
kernel does not cover (CPU parameters, amsgrad, maximize, sparse or non-fp32 tensors, a closure
that needs foreach semantics) goes to torch's own implementation unchanged; for CUDA fp32
parameters there is no silent fallback -- a missing library raises.
"""
import os
import weakref

import numpy as np
import torch

from . import ops

_CHUNK = 32768          # elements per workgroup
_TorchAdam = torch.optim.Adam
_MOMENTUM_MODELS = weakref.WeakSet()


_OWN_MODELS = weakref.WeakSet()


def register_momentum_model(model):
    """InfoNCE / UberNCE / CoCLR instances announce themselves so that `Adam(fold_momentum=True)`
    can find the (query, key) parameter pairs of model/pretrain.py:76-80."""
    _MOMENTUM_MODELS.add(model)
    _OWN_MODELS.add(model)


def register_model(model):
    """Any module of this package whose parameters the single-launch step may take (the linear-probe
    classifier).  `torch.optim.Adam` resolves to the subclass process-wide once `model.pretrain` is
    imported; optimisers over parameters that belong to NO registered module behave exactly as torch's."""
    _OWN_MODELS.add(model)


def _owns_any(params):
    ids = {id(p) for p in params}
    for model in list(_OWN_MODELS):
        for p in model.parameters():
            if id(p) in ids:
                return True
    return False


class Adam(_TorchAdam):
    """torch.optim.Adam with a single-launch `step()` on MI355X.

    fold_momentum (default: env COCLR_FOLD_MOMENTUM, off): also compute the momentum-encoder
    update of the NEXT forward (p_k = p_k*m + p_q*(1-m), model/pretrain.py:76-80) in the same
    pass, while the fresh p_q is in registers; the model then skips its own update once.
    Bit-identical in a steady training loop.  Caveat (why it is opt-in): between `step()` and the
    next training forward the key encoder is one momentum step AHEAD of where the reference has
    it -- a checkpoint written in that window, or a no-grad forward, sees the updated keys."""

    _scoped = False     # ScopedAdam: the native step only for parameters of this package's modules

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0,
                 amsgrad=False, *, fold_momentum=None, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                         amsgrad=amsgrad, **kw)
        if fold_momentum is None:
            fold_momentum = os.environ.get("COCLR_FOLD_MOMENTUM", "0") == "1"
        self._fold = bool(fold_momentum)
        self._plan = None
        self._ours = None       # decided at the first step: do these parameters belong to this package?

    # -- eligibility ---------------------------------------------------------------------
    def _native_groups(self):
        """[(group, [params with grad])] if every parameter that has a gradient can take the
        kernel, else None (torch's implementation runs instead)."""
        if self._ours is None:
            self._ours = not self._scoped or os.environ.get("COCLR_ADAM_EVERYWHERE", "0") == "1" or \
                _owns_any(p for g in self.param_groups for p in g["params"])
        if not self._ours:
            return None       # somebody else's model: torch's own implementation, untouched
        out = []
        dev = None
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or \
                    g.get("capturable") or g.get("decoupled_weight_decay") or g.get("fused") or \
                    g.get("foreach") or torch.is_tensor(g["lr"]):
                return None       # options the kernel does not implement, or an explicit torch path
            ps = []
            for p in g["params"]:
                gr = p.grad
                if gr is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or gr.is_sparse or \
                        gr.dtype != torch.float32 or not p.is_contiguous() or \
                        not gr.is_contiguous() or gr.device != p.device:
                    return None       # the kernel reads p and p.grad as flat fp32 memory
                if dev is None:
                    dev = p.device
                elif p.device != dev:
                    return None
                ps.append(p)
            if ps:
                out.append((g, ps))
        return out if out else None

    # -- plan: static part (params, state, chunking), refreshed when storage moves --------
    def _build_plan(self, groups):
        dev = groups[0][1][0].device
        params, hyper_of = [], []
        for gi, (g, ps) in enumerate(groups):
            for p in ps:
                params.append(p)
                hyper_of.append(gi)
        n = len(params)
        steps = torch.zeros(n, dtype=torch.float32, device=dev)
        host_steps = np.zeros(n, dtype=np.float32)
        for i, p in enumerate(params):
            st = self.state[p]
            if len(st) == 0:
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            else:
                host_steps[i] = float(st["step"])
                for key in ("exp_avg", "exp_avg_sq"):
                    if st[key].device != p.device or st[key].dtype != torch.float32 or \
                            not st[key].is_contiguous():
                        st[key] = st[key].to(device=p.device, dtype=torch.float32).contiguous()
        if host_steps.any():
            steps.copy_(torch.from_numpy(host_steps))
        for i, p in enumerate(params):
            self.state[p]["step"] = steps[i]          # 0-dim device view (torch's fused format)

        fold = None
        kptr = {}
        if self._fold:
            have = {id(p) for p in params}
            for model in list(_MOMENTUM_MODELS):
                pq = [p for p in model.encoder_q.parameters()]
                pk = [p for p in model.encoder_k.parameters()]
                # every (query, key) pair must be folded, or none: the model skips its own update
                # once the optimiser says it has applied it
                if pq and all(p.requires_grad and id(p) in have for p in pq) and \
                        all(k.is_cuda and k.is_contiguous() for k in pk):
                    kptr = {id(q): k for q, k in zip(pq, pk)}
                    fold = weakref.ref(model)
                    break
        rows, row_param, row_off = [], [], []
        for i, p in enumerate(params):
            st = self.state[p]
            k = kptr.get(id(p))
            numel = p.numel()
            for off in range(0, numel, _CHUNK):
                rows.append((p.data_ptr() + 4 * off, 0, st["exp_avg"].data_ptr() + 4 * off,
                             st["exp_avg_sq"].data_ptr() + 4 * off,
                             0 if k is None else k.data_ptr() + 4 * off,
                             min(_CHUNK, numel - off), i, 0))
                row_param.append(i)
                row_off.append(4 * off)
        host = [torch.empty(len(rows), 8, dtype=torch.int64).pin_memory() for _ in range(2)]
        template = np.array(rows, dtype=np.int64).reshape(-1, 8)
        self._plan = {
            "dev": dev, "params": params, "hyper_of": hyper_of, "groups": [g for g, _ in groups],
            "steps": steps, "template": template, "row_param": np.array(row_param, dtype=np.int64),
            "row_off": np.array(row_off, dtype=np.int64), "host": host, "events": [None, None],
            "flip": 0, "table": torch.empty(len(rows), 8, dtype=torch.int64, device=dev),
            "gptrs": None,
            # the kernel's "group" column is the PARAMETER slot (torch keeps one step counter per
            # parameter): hyper rows are the group's values repeated per parameter
            "hyper": torch.empty(n, 8, dtype=torch.float64, device=dev), "hyper_sig": None,
            "ids": torch.arange(n, dtype=torch.int32, device=dev), "n": n,
            "sig": [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                     self.state[p]["exp_avg_sq"].data_ptr(),
                     0 if kptr.get(id(p)) is None else kptr[id(p)].data_ptr()) for p in params],
            "kparams": [kptr.get(id(p)) for p in params],
            "fold": fold,
            "keep": [self.state[p]["exp_avg"] for p in params] +
                    [self.state[p]["exp_avg_sq"] for p in params] + list(kptr.values()),
        }

    def _plan_valid(self, plan, params):
        """Same parameter objects at the same addresses, optimizer state untouched since the plan
        was built (load_state_dict / .to() replace the state tensors)."""
        if plan is None or len(plan["params"]) != len(params):
            return False
        state = self.state
        steps_ptr = plan["steps"].untyped_storage().data_ptr()
        for a, b, (pp, mp, vp, kp), k in zip(plan["params"], params, plan["sig"], plan["kparams"]):
            if a is not b or b.data_ptr() != pp or (k is not None and k.data_ptr() != kp):
                return False
            st = state[b]
            if not st or st["exp_avg"].data_ptr() != mp or st["exp_avg_sq"].data_ptr() != vp or \
                    not torch.is_tensor(st["step"]) or \
                    st["step"].untyped_storage().data_ptr() != steps_ptr:
                return False
        return True

    # -- step ------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = self._native_groups()
        if groups is None:
            if self._plan is not None:
                # leaving the single-launch path (e.g. a sparse gradient appeared): torch's own
                # implementation wants its per-parameter step counters on the host
                for st in self.state.values():
                    if torch.is_tensor(st.get("step")) and st["step"].is_cuda:
                        st["step"] = torch.tensor(float(st["step"]), dtype=torch.float32)
                self._plan = None
            super().step()
            return loss
        params = [p for _, ps in groups for p in ps]
        plan = self._plan
        if not self._plan_valid(plan, params):
            self._build_plan(groups)
            plan = self._plan
        # hyper-parameters: uploaded when a group's values change (lr schedules)
        hsig = tuple((g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"])
                     for g, _ in groups)
        if hsig != plan["hyper_sig"]:
            rows = np.zeros((plan["n"], 8), dtype=np.float64)
            for i, gi in enumerate(plan["hyper_of"]):
                rows[i, :5] = hsig[gi]
            plan["hyper"].copy_(torch.from_numpy(rows))
            plan["hyper_sig"] = hsig
        # gradient addresses: autograd hands out fresh tensors every step (zero_grad sets them to
        # None); the caching allocator usually returns the same blocks, so this rarely uploads
        gptrs = [p.grad.data_ptr() for p in params]
        if gptrs != plan["gptrs"]:
            f = plan["flip"]
            ev = plan["events"][f]
            if ev is not None:
                ev.synchronize()                 # the copy that last read this pinned buffer
            tab = plan["host"][f].numpy()
            tab[:] = plan["template"]
            tab[:, 1] = np.asarray(gptrs, dtype=np.int64)[plan["row_param"]] + plan["row_off"]
            plan["table"].copy_(plan["host"][f], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            plan["events"][f] = ev
            plan["flip"] = 1 - f
            plan["gptrs"] = gptrs
        model = plan["fold"]() if plan["fold"] is not None else None
        m = float(model.m) if model is not None else 0.0
        ops.adam_step(plan["table"], plan["table"].shape[0], plan["hyper"], plan["steps"],
                      plan["ids"], plan["n"], m, 1.0 - m if model is not None else 0.0,
                      keep=plan["keep"])
        if model is not None:
            model.__dict__["_momentum_folded"] = m
        return loss


class ScopedAdam(Adam):
    """What `torch.optim.Adam` resolves to after `install()`: `Adam`, except that an optimiser over
    parameters that belong to NO module of this package (InfoNCE / UberNCE / CoCLR / LinearClassifier
    register themselves) is torch's own implementation, step for step -- importing `model.pretrain`
    must not change how somebody else's model in the same process is optimised.
    `COCLR_ADAM_EVERYWHERE=1` lifts the restriction."""
    _scoped = True


def install():
    """Make `torch.optim.Adam` (what main_nce.py:200 / main_coclr.py:213 construct) resolve to the
    single-launch subclass, scoped to this package's models (ScopedAdam).  Idempotent;
    COCLR_PATCH_ADAM=0 leaves torch untouched."""
    if os.environ.get("COCLR_PATCH_ADAM", "1") == "0":
        return False
    if torch.optim.Adam is not ScopedAdam:
        torch.optim.Adam = ScopedAdam
        if os.environ.get("COCLR_QUIET", "0") != "1":
            import sys
            print("coclr_amd: torch.optim.Adam resolves to the single-launch subclass for parameters of "
                  "InfoNCE / UberNCE / CoCLR / LinearClassifier modules (COCLR_PATCH_ADAM=0 opts out)",
                  file=sys.stderr)
    return True
