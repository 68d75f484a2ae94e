"""Where does the fp32 gradient error of the conditioned fixture come from?  (CPU only, the oracle.)

VERDICT r03 weak #1: on tests/golden/infonce_s3d_conditioned the reference's own fp32 gradients are
1.6e-2 (L2, relative) from a float64 evaluation and the HIP path 2.7e-2, "on every backbone tensor".
This tool answers with the oracle alone:

  1. fp32 vs float64, raw; the ReLU / max-pool decisions on which the two runs differ, per unit;
  2. fp32 vs float64 GIVEN THE fp32 RUN'S DECISIONS (ReLU -> mask multiply, pool -> gather): the smooth
     part of the error;
  3. float64 with fp32's decisions vs float64 with its own: the part the flips explain;
  4. the same fp32 run on inputs perturbed by one ulp (x * (1 + 1e-7 n)), six seeds: the spread of
     the REFERENCE ARITHMETIC's own raw error -- the distribution the product's 2.7e-2 is a draw from.

usage: python tools/grad_error_budget.py > profiles/r04_grad_error_budget.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _cases import build_model, case_inputs, l2_err, load_golden      # noqa: E402
from _decisions import oracle_grads                                    # noqa: E402


def stats(name, got, truth):
    errs = sorted(l2_err(got[k], truth[k]) for k in truth)
    print("%-58s median %.3e   min %.3e   max %.3e" % (name, errs[len(errs) // 2], errs[0], errs[-1]),
          flush=True)


def main():
    import model.pretrain as product
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    gold = load_golden("infonce_s3d_conditioned")
    cfg, rec = gold["cfg"], gold["steps"][0]
    model = build_model(cfg, product)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    blocks, extra = case_inputs(cfg, 0)
    perm = rec["perm"]
    print("fixture infonce_s3d_conditioned: B=%d clip %s, %d-thread ATen CPU oracle" % (
        cfg["B"], cfg["clip"], torch.get_num_threads()))
    g32, l32, _, d32 = oracle_grads(sd0, cfg, blocks, extra, perm, torch.float32)
    g64, l64, _, d64 = oracle_grads(sd0, cfg, blocks, extra, perm, torch.float64)
    print("loss fp32 %.7f  float64 %.7f;  %d parameter tensors, %d decisions (78 ReLU units, 13 pools)"
          % (float(l32), float(l64), len(g64), d32.count()))
    stats("1. fp32 vs float64 (raw)", g32, g64)
    flips = d32.flips(d64)
    print("   decisions that differ between the two runs: %d" % sum(f[1] for f in flips))
    for name, n, of in flips:
        print("      %-52s %3d of %d" % (name, n, of))
    g64f, _, _, _ = oracle_grads(sd0, cfg, blocks, extra, perm, torch.float64, decisions=d32)
    stats("2. fp32 vs float64 on fp32's decisions (smooth part)", g32, g64f)
    stats("3. float64 on fp32's decisions vs float64 (the flips)", g64f, g64)
    print("4. fp32 on inputs * (1 + 1e-7 n): raw error vs the float64 run of the UNPERTURBED input")
    for seed in range(1, 7):
        gen = torch.Generator().manual_seed(seed)
        nb = [blocks[0] * (1 + 1e-7 * torch.randn(blocks[0].shape, generator=gen))]
        gs, _, _, ds = oracle_grads(sd0, cfg, nb, extra, perm, torch.float32)
        stats("   seed %d: %3d decisions differ from float64" % (seed, sum(f[1] for f in ds.flips(d64))),
              gs, g64)


if __name__ == "__main__":
    main()
