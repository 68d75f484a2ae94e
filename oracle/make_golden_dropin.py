"""Generate tests/golden/dropin_main_nce.pt and dropin_main_coclr.pt: what the reference's OWN launch
scripts compute, with the reference's own model/backbone packages, over a few iterations on a
synthetic dataset -- the yardstick for running the same unmodified scripts on this repository's
shadow packages (tests/test_dropin_scripts.py) and for the restated caller loop on the GPU
(tests/test_gpu_dropin.py).

Run once in the build container:   python oracle/make_golden_dropin.py
Everything is imported unmodified from /root/reference through tests/dropin_harness.py (its
docstring lists the stand-ins an image without torchvision / tensorboardX / lmdb / a GPU needs).
"""
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import dropin_harness as H  # noqa: E402

CASES = {
    # name: (script, argv, dataset spec)
    "dropin_main_nce": ("main_nce", [
        "--net", "s3d", "--model", "infonce", "--moco-k", "32", "--batch_size", "4", "--seq_len", "16",
        "--img_dim", "64", "--epochs", "1", "--workers", "0", "--print_freq", "1", "--seed", "0",
        "--multiprocessing-distributed", "--world-size", "1", "--rank", "0", "--dist-backend", "gloo"],
        dict(n=12, seq_len=16, img_dim=64, two_stream=False, seed=77)),
    "dropin_main_coclr": ("main_coclr", [
        "--net", "s3d", "--model", "coclr", "--topk", "2", "--moco-k", "8", "--batch_size", "4",
        "--seq_len", "8", "--img_dim", "64", "--epochs", "1", "--workers", "0", "--print_freq", "1",
        "--seed", "0", "--multiprocessing-distributed", "--world-size", "1", "--rank", "0",
        "--dist-backend", "gloo", "--pretrain", "{tmp}/rgb.pth.tar", "{tmp}/flow.pth.tar"],
        dict(n=20, seq_len=8, img_dim=64, two_stream=True, seed=78)),
}
KEEP = ("queue", "queue_ptr", "queue_second", "queue_vname", "queue_label",
        "encoder_q.0.Conv_1a.bn1.running_mean", "encoder_q.0.Mixed_5c.branch0.0.bn.running_var",
        "encoder_k.0.Conv_2c.conv1.weight", "encoder_k.4.bias", "encoder_q.4.weight",
        "encoder_q.0.Conv_1a.conv1.weight")


def main():
    assert H.reference_available()
    out = os.path.join(ROOT, "tests", "golden")
    for name, (script, argv, spec) in CASES.items():
        ds = H.SyntheticClips(**spec)
        with tempfile.TemporaryDirectory() as tmp:
            if script == "main_coclr":
                H.write_pretrained_pair(tmp, use_reference_model=True)
            rec = H.run_script(script, [a.format(tmp=tmp) for a in argv], ds, use_reference_model=True,
                               cpu=True, workdir=tmp)
        sd = rec["checkpoint"]["state_dict"]
        opt = rec["checkpoint"]["optimizer"]
        gold = {
            "script": script, "argv": argv, "dataset": spec,
            "outputs": rec["outputs"], "targets": [t.nonzero() if t.dtype == torch.bool else t
                                                   for t in rec["targets"]],
            "losses": rec["losses"],
            "state_keys": list(sd.keys()),
            "state": {k: sd[k].clone() for k in KEEP if k in sd},
            "epoch": rec["checkpoint"]["epoch"], "iteration": rec["checkpoint"]["iteration"],
            "optimizer_groups": len(opt["param_groups"]),
            "optimizer_state_entries": len(opt["state"]),
            "made_by": "oracle/make_golden_dropin.py with /root/reference's main_*.py, model/, backbone/, "
                       "utils/ unmodified, torch %s CPU" % torch.__version__,
        }
        torch.save(gold, os.path.join(out, name + ".pt"))
        print(name, "iterations", len(rec["outputs"]), "losses", ["%.4f" % v for v in rec["losses"]],
              "groups", gold["optimizer_groups"], "state entries", gold["optimizer_state_entries"])


if __name__ == "__main__":
    main()
