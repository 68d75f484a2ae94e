"""Generate tests/golden/*.pt by running the UNMODIFIED reference (TengdaHan/CoCLR,
mounted read-only at /root/reference) on CPU.

Run once in the build container:   python oracle/make_golden.py
The GPU box has no /root/reference; tests only read the committed fixtures.

Harness (reference files untouched):
  * sys.path -> /root/reference so `model.pretrain` / `backbone.*` are the reference's
  * torch.Tensor.cuda -> identity  (model/pretrain.py:112,185 call .cuda() unconditionally)
  * gloo process group (the forward always calls collectives, pretrain.py:22,115)
  * torch.randperm is wrapped only to RECORD the permutation it returns

Every fixture records the seeds, the shapes, the outputs of step 1 and step 2
(Adam lr 1e-3 wd 1e-5 with one param group per tensor, as main_nce.py:190-200),
selected gradients, BatchNorm running statistics, queue contents and pointer.
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from _cases import condition_model  # noqa: E402  (shared with the tests that rebuild the model)

# a wider sample of backbone tensors for the well-conditioned fixture (every one is held per tensor)
GRAD_KEYS_S3D_WIDE = [
    "encoder_q.0.Conv_1a.conv1.weight", "encoder_q.0.Conv_1a.conv2.weight",
    "encoder_q.0.Conv_1a.bn1.weight", "encoder_q.0.Conv_1a.bn2.bias", "encoder_q.0.Conv_2b.conv.weight",
    "encoder_q.0.Conv_2c.conv1.weight", "encoder_q.0.Conv_2c.conv2.weight",
    "encoder_q.0.Conv_2c.bn2.weight", "encoder_q.0.Mixed_3b.branch0.0.conv.weight",
    "encoder_q.0.Mixed_3b.branch1.1.conv1.weight", "encoder_q.0.Mixed_3b.branch2.1.conv1.weight",
    "encoder_q.0.Mixed_3c.branch1.1.conv2.weight", "encoder_q.0.Mixed_3c.branch3.1.conv.weight",
    "encoder_q.0.Mixed_4b.branch1.0.conv.weight", "encoder_q.0.Mixed_4c.branch2.1.conv2.weight",
    "encoder_q.0.Mixed_4d.branch1.1.bn1.weight", "encoder_q.0.Mixed_4e.branch3.1.conv.weight",
    "encoder_q.0.Mixed_4f.branch1.1.conv1.weight", "encoder_q.0.Mixed_5b.branch0.0.bn.bias",
    "encoder_q.0.Mixed_5b.branch2.1.conv1.weight", "encoder_q.0.Mixed_5c.branch1.1.conv2.weight",
    "encoder_q.0.Mixed_5c.branch1.1.bn2.weight", "encoder_q.2.weight", "encoder_q.2.bias",
    "encoder_q.4.weight", "encoder_q.4.bias"]

GRAD_KEYS_S3D = ["encoder_q.0.Conv_1a.conv1.weight", "encoder_q.0.Conv_1a.bn1.weight",
                 "encoder_q.0.Conv_1a.bn1.bias", "encoder_q.0.Conv_2c.conv2.weight",
                 "encoder_q.0.Mixed_3b.branch2.1.conv1.weight",
                 "encoder_q.0.Mixed_4e.branch3.1.conv.weight",
                 "encoder_q.0.Mixed_5c.branch1.1.bn2.weight", "encoder_q.2.bias",
                 "encoder_q.4.weight", "encoder_q.4.bias"]
GRAD_KEYS_R50 = ["encoder_q.0.conv1.weight", "encoder_q.0.bn1.weight",
                 "encoder_q.0.layer1.0.downsample.0.weight", "encoder_q.0.layer2.0.conv2.weight",
                 "encoder_q.0.layer3.1.conv1.weight", "encoder_q.0.layer4.2.bn3.weight",
                 "encoder_q.4.bias"]
BUF_KEYS_S3D = ["encoder_q.0.Conv_1a.bn1.running_mean", "encoder_q.0.Conv_1a.bn1.running_var",
                "encoder_q.0.Mixed_5c.branch3.1.bn.running_mean",
                "encoder_q.0.Mixed_5c.branch3.1.bn.running_var",
                "encoder_k.0.Conv_1a.bn1.running_mean", "encoder_k.0.Mixed_4b.branch0.0.bn.running_var",
                "encoder_q.0.Conv_2b.bn.num_batches_tracked"]
BUF_KEYS_R50 = ["encoder_q.0.bn1.running_mean", "encoder_q.0.layer4.2.bn3.running_var",
                "encoder_k.0.layer2.0.downsample.1.running_mean"]
PARAM_KEYS = ["encoder_k.4.bias", "encoder_q.4.bias"]


def _setup_reference():
    sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    import model.pretrain as ref_pretrain
    assert ref_pretrain.__file__.startswith(REF), ref_pretrain.__file__
    return ref_pretrain


_PERMS = []
_orig_randperm = torch.randperm


def _recording_randperm(*a, **k):
    p = _orig_randperm(*a, **k)
    _PERMS.append(p.clone())
    return p


def _checksums(sd):
    """(keys, [n,2] float64 tensor of (sum, abs-sum)) -- compact on disk."""
    keys = [k for k, v in sd.items() if v.is_floating_point()]
    vals = torch.tensor([[float(sd[k].double().sum()), float(sd[k].double().abs().sum())]
                         for k in keys], dtype=torch.float64).reshape(len(keys), 2)
    return {"keys": keys, "vals": vals}


def _sample(t, limit=4096):
    """Whole tensor if small, else an evenly strided sample of `limit` elements."""
    flat = t.detach().reshape(-1)
    if flat.numel() <= limit:
        return flat.clone()
    step = flat.numel() // limit
    return flat[::step][:limit].clone()


def _loss(kind, out, tgt):
    if kind == "infonce":
        return F.cross_entropy(out, tgt)
    if kind == "ubernce":   # main_nce.py:320-321
        return (- (F.log_softmax(out, dim=1) * tgt).sum(1) / tgt.sum(1)).mean()
    return (- torch.log((F.softmax(out, dim=1) * tgt).sum(1))).mean()   # main_coclr.py:343-346


def run_case(ref, cfg, rank=0, world=1):
    kind, net = cfg["kind"], cfg["network"]
    B, K, dim = cfg["B"], cfg["K"], cfg["dim"]
    clip = cfg["clip"]                      # (C, T, H, W)
    torch.manual_seed(cfg["model_seed"])
    if kind == "infonce":
        model = ref.InfoNCE(net, dim, K, cfg["m"], cfg["T"])
        if cfg.get("condition"):
            condition_model(model, cfg["condition"])
    elif kind == "ubernce":
        model = ref.UberNCE(net, dim, K, cfg["m"], cfg["T"])
    else:
        model = ref.CoCLR(net, dim, K, cfg["m"], cfg["T"], topk=cfg["topk"],
                          reverse=cfg.get("reverse", False))
        if cfg.get("prefill"):
            # make the queue "full" with deterministic content so mining is active
            g = torch.Generator().manual_seed(cfg["prefill"])
            model.queue_label.fill_(1)
            model.queue_vname.copy_(torch.randint(0, cfg["n_sources"], (K,), generator=g))
    init_sums = _checksums(model.state_dict())
    wrapped = model
    if world > 1:
        wrapped = nn.parallel.DistributedDataParallel(model)
    params = [{"params": p} for _, p in wrapped.named_parameters()]
    opt = torch.optim.Adam(params, lr=1e-3, weight_decay=1e-5)
    model.train()
    if kind == "coclr":
        model.sampler.eval()               # main_coclr.py:363

    gold = {"cfg": cfg, "init_checksums": init_sums, "steps": []}
    grad_keys = GRAD_KEYS_R50 if net == "r50" else GRAD_KEYS_S3D
    if cfg.get("condition"):
        grad_keys = GRAD_KEYS_S3D_WIDE
    buf_keys = BUF_KEYS_R50 if net == "r50" else BUF_KEYS_S3D
    for step in range(cfg["steps"]):
        # every rank draws the whole global batch from the same seed and keeps its slice,
        # so a single-process test can rebuild all ranks' inputs
        g = torch.Generator().manual_seed(cfg["input_seed"] + step)
        nblk = 2 if kind == "coclr" else 1
        blocks = [torch.randn(B * world, 2, *clip, generator=g) for _ in range(nblk)]
        extra = None
        if kind == "ubernce":
            extra = torch.randint(0, cfg["n_classes"], (B * world,), generator=g)
        if kind == "coclr":
            extra = torch.randint(0, cfg["n_sources"], (B * world,), generator=g)
        sl = slice(rank * B, (rank + 1) * B)
        truth64 = None
        if cfg.get("truth64") and step == 0:
            # the REFERENCE model itself in float64 on the same weights, inputs and permutation: the
            # yardstick for how reproducible its own fp32 gradients are
            import copy
            m64 = copy.deepcopy(model).double()
            torch.manual_seed(cfg["perm_seed"] + step)
            o64, t64 = m64(blocks[0][sl].double())
            l64 = _loss(kind, o64, t64)
            l64.backward()
            n64 = dict(m64.named_parameters())
            truth64 = ({k: _sample(n64[k].grad) for k in grad_keys}, l64.detach().clone())
            del m64
        _PERMS.clear()
        torch.manual_seed(cfg["perm_seed"] + step)     # fixes randperm (pretrain.py:112)
        if kind == "infonce":
            out, tgt = wrapped(blocks[0][sl])
        elif kind == "ubernce":
            out, tgt = wrapped(blocks[0][sl], extra[sl])
        else:
            out, tgt = wrapped(blocks[0][sl], blocks[1][sl], extra[sl])
        loss = _loss(kind, out, tgt)
        opt.zero_grad()
        loss.backward()
        named = dict(model.named_parameters())
        rec = {
            "perm": _PERMS[0].clone(),
            "logits": out.detach().clone(),
            "target": tgt.detach().clone() if kind == "infonce" else tgt.nonzero().clone(),
            "loss": loss.detach().clone(),
            "grads": {k: _sample(named[k].grad) for k in grad_keys},
            "grad_checksums": _checksums({k: p.grad for k, p in named.items()
                                          if p.grad is not None}),
        }
        if truth64 is not None:
            rec["grads64"], rec["loss64"] = truth64
        opt.step()
        sd = model.state_dict()
        rec["buffers"] = {k: sd[k].detach().clone() for k in buf_keys}
        rec["params_after"] = {k: sd[k].detach().clone() for k in PARAM_KEYS}
        rec["queue_ptr"] = sd["queue_ptr"].clone()
        rec["queue_checksum"] = _checksums({"queue": sd["queue"]})["vals"][0]
        bw = B * world
        ptr0 = (int(sd["queue_ptr"]) - bw) % K
        rec["queue_cols"] = sd["queue"][:, ptr0:ptr0 + bw].clone()
        for extra_q in ("queue_label", "queue_vname"):
            if extra_q in sd:
                rec[extra_q + "_cols"] = sd[extra_q][ptr0:ptr0 + bw].clone()
        if "queue_second" in sd:
            rec["queue_second_cols"] = sd["queue_second"][:, ptr0:ptr0 + bw].clone()
        gold["steps"].append(rec)
    return gold


CASES = {
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    "infonce_s3d_config1": dict(kind="infonce", network="s3d", B=4, K=2048, dim=128, m=0.999,
                                T=0.07, clip=(3, 32, 128, 128), model_seed=0, input_seed=1,
                                perm_seed=100, steps=2),
    "infonce_s3d_small": dict(kind="infonce", network="s3d", B=4, K=32, dim=128, m=0.999, T=0.07,
                              clip=(3, 16, 64, 64), model_seed=0, input_seed=1, perm_seed=100,
                              steps=2),
    "ubernce_s3d_small": dict(kind="ubernce", network="s3d", B=4, K=32, dim=128, m=0.999, T=0.07,
                              clip=(3, 16, 64, 64), model_seed=2, input_seed=3, perm_seed=101,
                              n_classes=3, steps=2),
    "coclr_s3d_small": dict(kind="coclr", network="s3d", B=4, K=32, dim=128, m=0.999, T=0.07,
                            topk=5, clip=(3, 16, 64, 64), model_seed=4, input_seed=5,
                            perm_seed=102, n_sources=6, prefill=7, steps=2),
    "coclr_s3d_small_reverse_cold": dict(kind="coclr", network="s3d", B=4, K=32, dim=128, m=0.999,
                                         T=0.07, topk=5, reverse=True, clip=(3, 16, 64, 64),
                                         model_seed=4, input_seed=5, perm_seed=102, n_sources=6,
                                         steps=2),
    # He-initialised, key encoder perturbed: loss ~ log K, gradients O(1); float64 truth recorded
    "infonce_s3d_conditioned": dict(kind="infonce", network="s3d", B=4, K=32, dim=128, m=0.999, T=0.07,
                                    clip=(3, 16, 128, 128), model_seed=0, input_seed=11, perm_seed=110,
                                    steps=1, condition=dict(seed=9), truth64=True),
    "infonce_r50_small": dict(kind="infonce", network="r50", B=2, K=16, dim=128, m=0.999, T=0.07,
                              clip=(3, 8, 64, 64), model_seed=6, input_seed=7, perm_seed=103,
                              steps=2),
    "infonce_s3dg_small": dict(kind="infonce", network="s3dg", B=2, K=16, dim=128, m=0.999,
                               T=0.07, clip=(3, 16, 64, 64), model_seed=8, input_seed=9,
                               perm_seed=104, steps=1),
}
DDP_CASES = {
    "infonce_s3d_small_world2": dict(kind="infonce", network="s3d", B=2, K=32, dim=128, m=0.999,
                                     T=0.07, clip=(3, 16, 64, 64), model_seed=0, input_seed=1,
                                     perm_seed=100, steps=2),
    "coclr_s3d_small_world2": dict(kind="coclr", network="s3d", B=2, K=32, dim=128, m=0.999,
                                   T=0.07, topk=5, clip=(3, 16, 64, 64), model_seed=4,
                                   input_seed=5, perm_seed=102, n_sources=6, prefill=7, steps=2),
}


def _ddp_worker(rank, world, name, cfg, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(max(1, os.cpu_count() // world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ref = _setup_reference()
    torch.randperm = _recording_randperm
    gold = run_case(ref, cfg, rank=rank, world=world)
    torch.save(gold, os.path.join(OUT, "%s_rank%d.pt" % (name, rank)))
    dist.destroy_process_group()


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    ref = _setup_reference()
    torch.randperm = _recording_randperm
    for name, cfg in CASES.items():
        if only and name not in only:
            continue
        gold = run_case(ref, cfg)
        torch.save(gold, os.path.join(OUT, name + ".pt"))
        print(name, "loss", [float(s["loss"]) for s in gold["steps"]], flush=True)
    dist.destroy_process_group()
    for i, (name, cfg) in enumerate(DDP_CASES.items()):
        if only and name not in only:
            continue
        mp.spawn(_ddp_worker, args=(2, name, cfg, 29540 + i), nprocs=2, join=True)
        print(name, "done", flush=True)


if __name__ == "__main__":
    main()
