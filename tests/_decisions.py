"""Decision-conditioned gradient parity (test infrastructure).

d(loss)/d(w) of this network is a PIECEWISE-smooth function of the inputs: the pieces are the
ReLU decisions (78 units) and max-pool arg-max choices (13 pools) of the differentiated pass.  Two
correct fp32 evaluations that re-associate one sum flip a different handful of near-tie decisions
(~1e2 of 6e7 on the conditioned fixture) and then differ by percents on EVERY upstream tensor --
measured on the reference's own fp32 run against float64 (tools/grad_error_budget.py,
profiles/r04_grad_error_budget.txt): median L2 error 1.5e-2, of which 1.4e-4 remains once the float64
run is given the fp32 run's decisions.  So gradient parity is stated per piece:

    product gradients  vs  the oracle evaluated ON THE PRODUCT'S OWN DECISIONS

`record_product` collects the product's decisions through coclr_amd.engine.DECISION_PROBE;
`oracle_grads` runs the CPU oracle (fp32 or fp64), recording its decisions or forced onto given ones
(ReLU -> multiply by the mask, max-pool -> gather at the recorded arg-max)."""
import torch
import torch.nn.functional as F

from oracle import coclr_oracle as orc


class Decisions:
    def __init__(self):
        self.relu = {}       # BatchNorm unit name (reference state-dict prefix) -> bool mask
        self.pools = []      # arg-max (flat index in the (T,H,W) input volume), emission order

    def count(self):
        return sum(m.numel() for m in self.relu.values()) + sum(i.numel() for i in self.pools)

    def flips(self, other):
        """[(what, differing decisions, of)] against another set of decisions."""
        out = []
        for k, m in self.relu.items():
            d = int((m != other.relu[k]).sum())
            if d:
                out.append((k, d, m.numel()))
        for i, (a, b) in enumerate(zip(self.pools, other.pools)):
            d = int((a != b).sum())
            if d:
                out.append(("pool#%d" % i, d, a.numel()))
        return out


def record_product(model, fn):
    """Run fn() (one product forward of `model`) with the engine's decision probe installed."""
    from coclr_amd import engine
    names = {}
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            names.setdefault(id(m), name)
    dec = Decisions()

    def probe(kind, bn, t):
        if kind == "relu":
            name = bn if isinstance(bn, str) else names[id(bn)]
            if name in dec.relu:
                raise AssertionError("unit %s emitted twice in one differentiated pass" % name)
            dec.relu[name] = t.cpu()
        else:
            dec.pools.append(t.cpu().long())

    engine.DECISION_PROBE = probe
    try:
        out = fn()
    finally:
        engine.DECISION_PROBE = None
    return out, dec


class _FProxy:
    """torch.nn.functional as the oracle sees it while decisions are recorded or forced.  Only the
    differentiated (query-encoder) pass is touched: the key / sampler passes run under no_grad."""

    def __init__(self, force):
        self.force = force
        self.dec = Decisions()
        self.cur = None
        self.ip = 0
        # force mode: decisions the oracle's OWN arithmetic would have taken differently at this unit, given the
        # same (forced) decisions everywhere upstream -- [(what, differing, of)]
        self.disagree = []

    def __getattr__(self, name):
        return getattr(F, name)

    def relu(self, x):
        if not x.requires_grad:
            return F.relu(x)
        # a ReLU that does not follow a BatchNorm is the projection head's (model/pretrain.py:53)
        name, self.cur = (self.cur or "head"), None
        if self.force is not None:
            m = self.force.relu[name]
            d = int(((x.detach() > 0) != m).sum())
            if d:
                self.disagree.append((name, d, m.numel()))
            return x * m.to(x.dtype)
        self.dec.relu[name] = x.detach() > 0
        return F.relu(x)

    def max_pool3d(self, x, k, s=None, p=0):
        if not x.requires_grad:
            return F.max_pool3d(x, k, s, p)
        if self.force is not None:
            idx = self.force.pools[self.ip]
            self.ip += 1
            n, c = x.shape[:2]
            picked = x.reshape(n, c, -1).gather(2, idx.reshape(n, c, -1)).reshape(idx.shape)
            # a different arg-max only counts when it selects a different VALUE (ties are the same decision)
            own = F.max_pool3d(x.detach(), k, s, p)
            d = int((own != picked.detach()).sum())
            if d:
                self.disagree.append(("pool#%d" % (self.ip - 1), d, idx.numel()))
            return picked
        out, idx = F.max_pool3d(x, k, s, p, return_indices=True)
        self.dec.pools.append(idx)
        return out


def oracle_grads(state_dict, cfg, blocks, extra, perm, dtype=torch.float32, decisions=None):
    """One oracle step from `state_dict` (CPU tensors, reference keys).  Returns
    ({parameter name: gradient} over encoder_q, without S3D's alias keys; loss; logits; decisions)."""
    kind = cfg["kind"]
    sd = orc.training_state({k: v.detach().cpu() for k, v in state_dict.items()})
    if dtype == torch.float64:
        sd = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.is_floating_point() else v)
              for k, v in sd.items()}
    pb = [blocks[0].to(dtype)] if kind != "coclr" else [(blocks[0].to(dtype), blocks[1].to(dtype))]
    proxy = _FProxy(decisions)
    real_bn = orc._bn

    def bn(sd_, pre, x, training):
        # the ReLU of a ResNet bottleneck follows bn3 + residual (backbone/resnet_2d3d.py:82-86): the
        # BatchNorm of the downsample path in between (:79-80) is not the unit it belongs to
        if x.requires_grad and ".downsample." not in pre:
            proxy.cur = pre
        return real_bn(sd_, pre, x, training)

    orc.F, orc._bn = proxy, bn
    try:
        outs = orc.nce_step(sd, kind, cfg["network"], pb, [extra], cfg["dim"], cfg["K"], cfg["m"],
                            cfg["T"], perm, topk=cfg.get("topk", 5), reverse=cfg.get("reverse", False))
    finally:
        orc.F, orc._bn = F, real_bn
    logits, tgt = outs[0]
    from _cases import loss_fn
    loss = loss_fn(kind, logits, tgt)
    loss.backward()
    grads, seen = {}, set()
    for k, v in sd.items():
        if not (torch.is_tensor(v) and v.requires_grad and v.grad is not None):
            continue
        if v.data_ptr() in seen or ".block" in k:
            continue
        seen.add(v.data_ptr())
        grads[k] = v.grad
    proxy.dec.disagree = proxy.disagree
    return grads, loss.detach(), logits.detach(), proxy.dec


def product_grads(model):
    """{reference key: .grad} of the product model, same key set as oracle_grads."""
    grads, seen = {}, set()
    for k, p in model.named_parameters():
        if p.grad is None or ".block" in k or id(p) in seen or not k.startswith("encoder_q."):
            continue
        seen.add(id(p))
        grads[k] = p.grad.detach().cpu()
    return grads


def l2_table(got, ref, truth):
    """{key: (L2 error of got vs truth, L2 error of ref vs truth)} relative to |truth|."""
    from _cases import l2_err
    return {k: (l2_err(got[k], truth[k]), l2_err(ref[k], truth[k])) for k in truth if k in got}
