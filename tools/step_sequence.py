"""Ordered kernel sequence of ONE steady-state step from a rocprofv3 kernel trace of a serialised bench run
(COCLR_OVERLAP_KEYS=0 COCLR_WGRAD_STREAM=0 COCLR_GRAPHS=0): start offset, duration, gap before, grid,
registers, name.  Segments are cut at the stage max-pools so the stage totals (kernel time AND gaps) can be
read off.  usage: python tools/step_sequence.py <kernel_trace.csv> [out.txt]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", name)
    gx = int(r.get("Grid_Size_X", 0) or 0) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
    wg = int(r.get("Workgroup_Size_X", 1) or 1) * int(r.get("Workgroup_Size_Y", 1) or 1) * int(r.get("Workgroup_Size_Z", 1) or 1)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, gx // max(wg, 1), wg,
               int(r.get("VGPR_Count", 0) or 0) + int(r.get("Accum_VGPR_Count", 0) or 0),
               int(r.get("LDS_Block_Size", 0) or 0)))
ev.sort()
ends = [i for i, e in enumerate(ev) if "adam_multi_kernel" in e[2]]
if len(ends) < 2:
    raise SystemExit("need at least two optimiser steps in the trace")
lo, hi = ends[-2] + 1, ends[-1] + 1
step = ev[lo:hi]
t0 = step[0][0]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("# one step: %d launches, wall %.3f ms, kernel time %.3f ms" % (
    len(step), (step[-1][1] - t0) / 1e6, sum(e - s for s, e, *_ in step) / 1e6), file=out)
print("# %9s %8s %7s %6s %5s %4s %6s  %s" % ("start us", "dur us", "gap us", "WGs", "thr", "regs", "LDS", "kernel"), file=out)
seg_k = seg_g = 0.0
seg_n = 0
prev_end = t0
segs = []
for s, e, name, wgs, wg, regs, lds in step:
    gap = max(0, s - prev_end)
    if "maxpool3d_tiled" in name and seg_n:
        segs.append((seg_n, seg_k, seg_g))
        print("# ---- segment: %d launches, kernels %.1f us, gaps %.1f us" % (seg_n, seg_k / 1e3, seg_g / 1e3), file=out)
        seg_k = seg_g = 0.0
        seg_n = 0
    print("%11.1f %8.1f %7.1f %6d %5d %4d %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, wgs, wg, regs, lds,
                                                     name[:120]), file=out)
    seg_k += e - s
    seg_g += gap
    seg_n += 1
    prev_end = max(prev_end, e)
segs.append((seg_n, seg_k, seg_g))
print("# ---- segment: %d launches, kernels %.1f us, gaps %.1f us" % (seg_n, seg_k / 1e3, seg_g / 1e3), file=out)
print("# segments (cut at the strided max-pools, forward then backward): launches / kernel ms / gap ms", file=out)
for n, k, g in segs:
    print("#   %4d  %7.3f  %7.3f" % (n, k / 1e6, g / 1e6), file=out)
