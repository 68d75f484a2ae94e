"""Fresh-process run of the main_coclr.py call sequence (tests/_caller_loop.py) on the GPU; prints the
relative error of each iteration's logits against the fixture recorded from the reference's scripts."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29612")
dist.init_process_group("nccl", rank=0, world_size=1)
import dropin_harness as H, _caller_loop
import model.pretrain as product
from oracle import coclr_oracle as orc
from _cases import load_golden, rel_err
gold = load_golden("dropin_main_coclr")
ds = H.SyntheticClips(**gold["dataset"])
with tempfile.TemporaryDirectory() as tmp:
    H.write_pretrained_pair(tmp, use_reference_model=False, product=product)
    rec = _caller_loop.run_coclr(product, ds, gpu=0, calc_topk_accuracy=orc.calc_topk_accuracy,
                                 calc_mask_accuracy=orc.calc_mask_accuracy,
                                 pretrain=(os.path.join(tmp, "rgb.pth.tar"), os.path.join(tmp, "flow.pth.tar")))
print("RESULT", os.environ.get("TAG", ""), " ".join("%.2e" % rel_err(a, b) for a, b in zip(rec["outputs"], gold["outputs"])))
if os.environ.get("SAVE"):
    torch.save(rec["outputs"], os.environ["SAVE"])
if os.environ.get("TWICE"):
    with tempfile.TemporaryDirectory() as tmp:
        H.write_pretrained_pair(tmp, use_reference_model=False, product=product)
        rec2 = _caller_loop.run_coclr(product, ds, gpu=0, calc_topk_accuracy=orc.calc_topk_accuracy,
                                      calc_mask_accuracy=orc.calc_mask_accuracy,
                                      pretrain=(os.path.join(tmp, "rgb.pth.tar"), os.path.join(tmp, "flow.pth.tar")))
    print("RESULT second run in the same process",
          " ".join("%.2e" % rel_err(a, b) for a, b in zip(rec2["outputs"], gold["outputs"])))
dist.destroy_process_group()
