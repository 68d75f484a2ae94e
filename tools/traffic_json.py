"""Fold a tools/pmc_layers.sh summary (gpurun_out/r0N_pmc_dominant.txt) into profiles/r0N_traffic.json:
HBM bytes per launch of the spatial Winograd kernel on Conv_2c.conv1 (FETCH_SIZE / WRITE_SIZE are in
KiB-like units of 1000 B as rocprofv3 prints them; FETCH doubled as MI355X_MICROARCH.md prescribes for
this rocprofv3 on gfx950) and its MFMA-busy fraction."""
import hashlib, json, os, re, sys
src, dst = sys.argv[1], sys.argv[2]


def csrc_sha16():
    """Hash of the kernel sources the counters were collected on: bench.py refuses the file for any other build."""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "coclr_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]



txt = open(src).read()
key = "void conv_wino_hw8_kernel" if "void conv_wino_hw8_kernel" in txt else "void conv_wino_hw_kernel"
blk = txt[txt.index(key):]
blk = blk[:blk.index("\nvoid ", 5)] if "\nvoid " in blk[5:] else blk
name = blk.splitlines()[0].strip()
val = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=", blk, re.M)}
fetch, write = val["FETCH_SIZE"] * 1000.0 * 2.0, val["WRITE_SIZE"] * 1000.0
alg = 4.0 * 32 * 16 * 32 * 32 * (64 + 192)           # x + y of one launch (weights are L2 hits)
cycles_per_xcd = val["GRBM_GUI_ACTIVE"] / 8.0
busy = val["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cycles_per_xcd)
out = {"csrc_sha16": csrc_sha16(), "kernel": name + " Winograd F(2x2,3x3) (Conv_2c.conv1 fwd 64->192 and dgrad 192->64, N=32, 16x32x32)",
       "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes, "
                 "tools/pmc_layers.sh Conv_2c.conv1; FETCH_SIZE doubled as MI355X_MICROARCH.md "
                 "prescribes for wide streaming reads on gfx950)" % src.split("/")[-1],
       "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "algorithmic_bytes_per_launch": alg,
       "mfma_busy_frac": round(busy, 4),
       "lds_bank_conflict_cycles": val.get("SQ_LDS_BANK_CONFLICT"),
       "note": "mean over forward + data-gradient launches (same kernel); MFMA busy = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)"}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out))
