"""Generate tests/golden/next_*.pt: reference outputs for the loop neighbours of SURVEY.md 8f
(loss + accuracy epilogue, input staging, optimiser step, LinearClassifier, NN retrieval).

Run once in the build container:   python oracle/make_golden_next.py
Reference files are imported UNMODIFIED from /root/reference.  Harness:
  * `torchvision` is absent from this image and utils/utils.py:7, utils/transforms.py:9 import it at
    module level (the functions used here never touch it): an empty stand-in module is put in
    sys.modules for the import only;
  * main_coclr.py cannot be imported (tensorboardX, lmdb ...): the source text of its
    `multi_nce_loss` function (main_coclr.py:343-346) is exec'd from the file as it lies there;
  * the UberNCE loss (main_nce.py:322-323) and `tr` (main_nce.py:299-302) are expressions inside
    train_one_epoch, restated here in the reference's own words;
  * utils/utils.py:67 calls `.view(-1)` on a non-contiguous slice, which PyTorch >= 1.7 rejects (the
    reference pins PyTorch 1.4): Tensor.view falls back to reshape for that call only;
  * the retrieval block (eval/main_classifier.py:686-706) is inline script code: it is cut out of
    the file by line range and exec'd (`_reference_retrieval_block`; fixture marked `pinned: True`).
"""
import ast
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


def _reference_function(path, name, env):
    src = open(os.path.join(REF, path)).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module([node], []), os.path.join(REF, path), "exec")
    exec(code, env)
    return env[name]


def main():
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    import utils.utils as ref_utils
    import utils.transforms as ref_T
    assert ref_utils.__file__.startswith(REF) and ref_T.__file__.startswith(REF)
    multi_nce_loss = _reference_function("main_coclr.py", "multi_nce_loss",
                                         {"torch": torch, "F": F})
    os.makedirs(OUT, exist_ok=True)

    # ---- loss + accuracy epilogue ---------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    gold = {"cases": []}
    for (B, K) in [(6, 300), (32, 2048)]:
        q = F.normalize(torch.randn(B, 128, generator=g), dim=1)
        queue = F.normalize(torch.randn(128, K, generator=g), dim=0)
        k = F.normalize(q + 0.5 * torch.randn(B, 128, generator=g), dim=1)
        logits = torch.cat([(q * k).sum(1, keepdim=True), q @ queue], 1) / 0.07
        mask = torch.rand(B, 1 + K, generator=g) < (3.0 / K)
        mask[:, 0] = True
        mask[1, 1:] = False                      # a row whose only positive is itself
        target = torch.zeros(B, dtype=torch.long)
        rec = {"logits": logits.clone(), "mask": mask.clone(), "target": target}
        for name, fn in (
                ("ce", lambda lg: nn.CrossEntropyLoss()(lg, target)),                  # main_nce.py:201,315
                ("multi", lambda lg: multi_nce_loss(lg, mask)),                        # main_coclr.py:389
                ("multi_drop", lambda lg: multi_nce_loss(lg, _drop_self(mask))),       # main_coclr.py:384-387
                ("uber", lambda lg: (- (F.log_softmax(lg, dim=1) * mask).sum(1) / mask.sum(1)).mean())):  # main_nce.py:322-323
            lg = logits.clone().requires_grad_(True)
            loss = fn(lg)
            loss.backward()
            rec[name] = {"loss": loss.detach().clone(), "dlogits": lg.grad.clone()}
        _view = torch.Tensor.view

        def _lenient_view(self, *a, **k):
            try:
                return _view(self, *a, **k)
            except RuntimeError:
                return self.reshape(*a, **k)
        torch.Tensor.view = _lenient_view
        try:
            rec["topk_self"] = [t.clone() for t in ref_utils.calc_topk_accuracy(logits, target, (1, 5))]
        finally:
            torch.Tensor.view = _view
        rec["topk_mask"] = [t.clone() for t in ref_utils.calc_mask_accuracy(logits, mask, (1, 5))]
        gold["cases"].append(rec)
    torch.save(gold, os.path.join(OUT, "next_loss_epilogue.pt"))
    print("loss epilogue:", [float(c["multi"]["loss"]) for c in gold["cases"]])

    # ---- input staging ----------------------------------------------------------------------
    g = torch.Generator().manual_seed(12)
    B, S, T, H = 3, 2, 4, 16
    u8 = torch.randint(0, 256, (B, 3, S * T, H, H), generator=g, dtype=torch.uint8)
    f32 = u8.to(torch.float32) / 255          # torchvision ToTensor (utils/transforms.py:49-51 form)
    norm = ref_T.Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], channel=1)
    out = norm(f32).view(B, 3, S, T, H, H).transpose(1, 2).contiguous()      # main_nce.py:299-302
    torch.save({"u8": u8, "num_seq": S, "seq_len": T, "out": out}, os.path.join(OUT, "next_staging.pt"))
    print("staging:", float(out.abs().sum()))

    # ---- optimiser: torch.optim.Adam over one group per tensor (main_nce.py:190-200) -----------
    g = torch.Generator().manual_seed(13)
    shapes = [(40000,), (33,), (48, 16, 1, 3, 3), (5,)]
    p0 = [torch.randn(*s, generator=g) * 0.1 for s in shapes]
    grads = [[torch.randn(*s, generator=g) * 0.01 for s in shapes] for _ in range(3)]
    ps = [p.clone().requires_grad_(True) for p in p0]
    opt = torch.optim.Adam([{"params": p} for p in ps], lr=1e-3, weight_decay=1e-5)
    traj = []
    for step in range(3):
        for p, gr in zip(ps, grads[step]):
            p.grad = gr.clone()
        opt.step()
        traj.append([p.detach().clone() for p in ps])
    torch.save({"p0": p0, "grads": grads, "after": traj, "lr": 1e-3, "wd": 1e-5,
                "betas": (0.9, 0.999), "eps": 1e-8}, os.path.join(OUT, "next_adam.pt"))
    print("adam:", float(traj[-1][0].double().sum()))

    # ---- LinearClassifier (model/classifier.py) -------------------------------------------------
    import model.classifier as ref_cls
    assert ref_cls.__file__.startswith(REF)
    torch.manual_seed(21)
    clf = ref_cls.LinearClassifier(num_class=7, network='s3d', dropout=0.5, use_dropout=True,
                                   use_l2_norm=True, use_final_bn=True)
    init = {k: v.clone() for k, v in clf.state_dict().items()}
    g = torch.Generator().manual_seed(22)
    block = torch.randn(4, 3, 16, 64, 64, generator=g)     # rebuilt from the seed by the tests
    clf.eval()
    with torch.no_grad():
        logit_eval, feat_eval = clf(block)
    clf.train()
    clf.final_fc[0].p = 0.0                     # dropout noise off: the rest of the train-mode path
    logit_tr, feat_tr = clf(block)
    tgt = torch.tensor([1, 4, 6, 0])
    F.cross_entropy(logit_tr, tgt).backward()
    named = dict(clf.named_parameters())
    keys = ["final_fc.1.weight", "final_fc.1.bias", "final_bn.weight", "backbone.Conv_1a.conv1.weight",
            "backbone.Mixed_5c.branch0.0.conv.weight"]
    sd = clf.state_dict()
    torch.save({"cfg": dict(num_class=7, network='s3d', use_l2_norm=True, use_final_bn=True),
                "init_keys": sorted(init), "block_seed": 22, "block_shape": (4, 3, 16, 64, 64),
                "target": tgt,
                "init_fc_weight": init["final_fc.1.weight"], "init_sum": float(sum(
                    v.double().sum() for v in init.values() if v.is_floating_point())),
                "logit_eval": logit_eval, "feat_eval": feat_eval,
                "logit_train": logit_tr.detach(), "feat_train": feat_tr.detach(),
                "grads": {k: named[k].grad.reshape(-1)[:4096].clone() for k in keys},
                "final_bn.running_mean": sd["final_bn.running_mean"].clone(),
                "final_bn.running_var": sd["final_bn.running_var"].clone()},
               os.path.join(OUT, "next_classifier.pt"))
    print("classifier:", logit_eval[0, :3].tolist())

    # ---- NN retrieval: the inline block eval/main_classifier.py:686-706, exec'd by line range -------
    sys.path.insert(0, os.path.join(HERE, ".."))
    from oracle import coclr_oracle as orc
    g = torch.Generator().manual_seed(31)
    ntr, nte, Cc, ncls = 700, 90, 64, 25
    centers = torch.randn(ncls, Cc, generator=g)
    trl = torch.randint(0, ncls, (ntr,), generator=g)
    tel = torch.randint(0, ncls, (nte,), generator=g)
    trf = centers[trl] * 0.3 + torch.randn(ntr, Cc, generator=g) + 2.0
    tef = centers[tel] * 0.3 + torch.randn(nte, Cc, generator=g) + 2.0
    acc, sim = _reference_retrieval_block(tef.clone(), tel, trf.clone(), trl)
    acc_o, sim_o = orc.nn_retrieval(tef, tel, trf, trl)           # the restatement agrees
    assert acc == acc_o and torch.equal(sim, sim_o)
    torch.save({"pinned": True, "train_feature": trf, "train_label": trl, "test_feature": tef,
                "test_label": tel, "acc": acc, "sim": sim}, os.path.join(OUT, "next_retrieval.pt"))
    print("retrieval:", acc)


def _reference_retrieval_block(test_feature, test_label, train_feature, train_label):
    """Run eval/main_classifier.py lines 686-706 (`ks = [1,5,10,20,50]` ... the k-NN loop's print) as
    they lie in the file: the block is inline code of `test_retrieval`, so it is cut out by line range,
    dedented and exec'd with the four tensors it reads in scope.  `torch.save(sim, ...)` inside it
    writes into a scratch directory."""
    import tempfile
    import textwrap
    import types
    lines = open(os.path.join(REF, "eval", "main_classifier.py")).read().split("\n")
    first = next(i for i, l in enumerate(lines) if l.strip() == "ks = [1,5,10,20,50]")
    last = next(i for i in range(first, len(lines)) if "print('%dNN acc = %.4f' % (k, acc))" in lines[i])
    assert (first + 1, last + 1) == (686, 706), (first + 1, last + 1)
    src = textwrap.dedent("\n".join(lines[first:last + 1]))
    with tempfile.TemporaryDirectory() as tmp:
        env = {"torch": torch, "F": F, "os": os, "dirname": "",
               "args": types.SimpleNamespace(test=os.path.join(tmp, "ckpt.pth.tar"), dataset="synthetic"),
               "test_feature": test_feature, "train_feature": train_feature,
               "test_label": test_label, "train_label": train_label}
        exec(compile(src, os.path.join(REF, "eval", "main_classifier.py"), "exec"), env)
    return env["NN_acc"], env["sim"]


def _drop_self(mask):
    mask_clone = mask.clone()                   # main_coclr.py:384-387
    mask_clone[mask.sum(1) != 1, 0] = 0
    return mask_clone


if __name__ == "__main__":
    main()
