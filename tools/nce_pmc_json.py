"""Fold a tools/pmc_layers.sh summary of tools/bench_nce.py (KS=16384) into profiles/r0N_nce_pmc.json: MFMA-busy
fraction, achieved FLOP/s and bytes/s of the q.queue^T logits kernel (model/pretrain.py:175-182).
usage: python tools/nce_pmc_json.py <pmc summary txt> <kernel trace csv dir> <out json>"""
import csv
import glob
import json
import os
import re
import sys

src, tracedir, dst = sys.argv[1:4]
txt = open(src).read()
blk = txt[txt.index("nce_logits_kernel"):]
end = re.search(r"\n\S", blk[5:])
blk = blk[:end.start() + 5] if end else blk
val = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=", blk, re.M)}
durs = []
for f in glob.glob(os.path.join(tracedir, "p1", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "nce_logits_kernel" in r["Kernel_Name"]:
            durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
durs.sort()
us = durs[len(durs) // 2] / 1e3 if durs else None
B, D, K = 32, 128, 16384
flop = 2.0 * B * D * K
byts = 4.0 * (D * K + 2 * B * D + B * (1 + K))
# SIMD cycles from the kernel's traced duration at the 2.2 GHz the chip sustains (as in r02): GRBM_GUI_ACTIVE of a
# 7-8 us launch under a counter pass is dominated by dispatch overhead
simd_cycles = 1024.0 * (us or 0.0) * 1e-6 * 2.2e9
busy = val["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles if simd_cycles else 0.0
out = {"source": "profiles/%s (rocprofv3 --pmc, separate passes, tools/pmc_layers.sh with PMC_SCRIPT=tools/bench_nce.py "
                 "KS=16384; kernel duration = median of the same runs' kernel trace)" % os.path.basename(src),
       "kernel": "nce_logits_kernel<128>, B=32, K=16384 launch (256 workgroups)",
       "kernel_duration_us": us, "mfma_busy_frac": round(busy, 4),
       "mfma_busy_cycles_per_launch": val["SQ_VALU_MFMA_BUSY_CYCLES"], "simd_cycles_per_launch": round(simd_cycles),
       "tflops": round(flop / us / 1e6, 2) if us else None,
       "frac_of_fp32_mfma_peak": round(flop / us / 1e6 / 157.3, 4) if us else None,
       "algorithmic_gbs": round(byts / us / 1e3, 1) if us else None,
       "frac_of_hbm_peak": round(byts / us / 1e3 / 8000.0, 4) if us else None,
       "fetch_kb_per_launch": val.get("FETCH_SIZE"), "write_kb_per_launch": val.get("WRITE_SIZE"),
       "bound_from_survey_8d": "<= 0.51-0.65 of the fp32 MFMA peak at perfect HBM bandwidth (12.8 FLOP/B); at 10.5 MB "
                               "per launch the kernel is launch/latency bound (same duration at K=2048 and K=16384)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
