"""Fold the rocprofv3 --pmc passes of tools/pmc_step.sh over ONE steady-state training step (the dispatches between
the last two optimiser launches): per kernel -- launches, GPU time (GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz: the counter
passes serialise the dispatches), MFMA-busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)), MFMA FLOPs
ISSUED (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512), HBM fetch (FETCH_SIZE x 1000 B, doubled: the gfx950 correction of
MI355X_MICROARCH.md for wide streaming reads) and write bytes.  Writes the table and a JSON with the step totals
that bench.py turns into `step_roofline.mfma_issued_frac` (stamped with a hash of the kernel sources: bench.py
refuses it for any other build).
usage: python tools/pmc_step_table.py <dir with p1 p2 p3> <table.txt> <step.json>"""
import csv
import glob
import hashlib
import json
import os
import re
import sys

root, table_path, json_path = sys.argv[1], sys.argv[2], sys.argv[3]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(REPO, "coclr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)[:100]


per_kernel = {}      # name -> {counter: sum over the step's dispatches}, "n": launches
for p in sorted(glob.glob(os.path.join(root, "p*"))):
    if not os.path.isdir(p):
        continue
    files = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    disp = {}
    for r in rows:
        d = disp.setdefault(int(r["Dispatch_Id"]), {"name": short(r["Kernel_Name"]), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(disp)
    ends = [i for i in ids if "adam_multi_kernel" in disp[i]["name"]]
    if len(ends) < 2:
        raise SystemExit("need two optimiser launches in %s" % p)
    step = [i for i in ids if ends[-2] < i <= ends[-1]]
    seen_n = {}
    for i in step:
        d = disp[i]
        k = per_kernel.setdefault(d["name"], {})
        seen_n[d["name"]] = seen_n.get(d["name"], 0) + 1
        for c, v in d["c"].items():
            k[c] = k.get(c, 0.0) + v
    for name, n in seen_n.items():
        per_kernel[name]["n"] = n

CLOCK = 2.4e9
tot_cycles = sum(k.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 for k in per_kernel.values())
tot_flop = sum(k.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 for k in per_kernel.values())
tot_fetch = sum(k.get("FETCH_SIZE", 0.0) * 1000.0 * 2.0 for k in per_kernel.values())
tot_write = sum(k.get("WRITE_SIZE", 0.0) * 1000.0 for k in per_kernel.values())
lines = ["# one steady-state step of bench.py (S3D InfoNCE, B=32, K=2048) under rocprofv3 --pmc, serialised; "
         "csrc %s" % csrc_sha16(),
         "# time = GRBM_GUI_ACTIVE / 8 / 2.4 GHz; MFMA %% = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x cycles); issued = "
         "SQ_INSTS_VALU_MFMA_MOPS_F32 x 512; fetch = 2 x FETCH_SIZE x 1000 B (gfx950 correction)",
         "# step totals: %.2f ms of kernel time, %.1f GFLOP issued on the matrix pipes (%.3f of the fp32 MFMA "
         "peak over that time), fetch %.0f MB, write %.0f MB"
         % (tot_cycles / CLOCK * 1e3, tot_flop / 1e9, tot_flop / (tot_cycles / CLOCK) / 157.3e12,
            tot_fetch / 1e6, tot_write / 1e6),
         "%-84s %5s %8s %6s %6s %9s %9s %9s" % ("kernel", "n", "ms/step", "% step", "MFMA %", "issued GF",
                                                "fetch MB", "write MB")]
order = sorted(per_kernel.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))
for name, k in order:
    cyc = k.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc < 0.002 * tot_cycles:
        continue
    busy = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc) if cyc else 0.0
    lines.append("%-84s %5d %8.3f %6.1f %6.1f %9.1f %9.1f %9.1f" % (
        name[:84], k.get("n", 0), cyc / CLOCK * 1e3, 100.0 * cyc / tot_cycles, 100.0 * busy,
        k.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) * 512.0 / 1e9, k.get("FETCH_SIZE", 0.0) * 2e3 / 1e6,
        k.get("WRITE_SIZE", 0.0) * 1e3 / 1e6))
open(table_path, "w").write("\n".join(lines) + "\n")
dom = per_kernel.get(next((n for n in per_kernel if n.startswith("conv_wino_hw8_kernel")), ""), None)
out = {"csrc_sha16": csrc_sha16(), "workload": "s3d infonce B=32 K=2048 3x32x128x128, one training step",
       "mfma_issued_gflop_per_step": round(tot_flop / 1e9, 2),
       "hbm_fetch_mb_per_step": round(tot_fetch / 1e6, 1), "hbm_write_mb_per_step": round(tot_write / 1e6, 1),
       "serialised_kernel_ms_per_step": round(tot_cycles / CLOCK * 1e3, 3),
       "source": "profiles/%s (tools/pmc_step.sh: rocprofv3 --pmc, separate passes for SQ_*, FETCH_SIZE, "
                 "WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes on gfx950)"
                 % os.path.basename(table_path)}
if dom is not None and dom.get("n"):
    n = dom["n"]
    out["dominant_kernel"] = {
        "kernel": "conv_wino_hw8_kernel (all %d launches of one step: forward and data gradient of the "
                  ">=16x16 (1,3,3) layers)" % n,
        "fetch_bytes_per_step": dom.get("FETCH_SIZE", 0.0) * 2e3, "write_bytes_per_step": dom.get("WRITE_SIZE", 0.0) * 1e3,
        "mfma_busy_frac": round(dom.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * dom["GRBM_GUI_ACTIVE"] / 8.0), 4)}
json.dump(out, open(json_path, "w"), indent=1)
print("\n".join(lines[:40]))
print(json.dumps(out))
