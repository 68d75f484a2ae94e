"""Timeline view of a rocprofv3 kernel trace of bench.py: per step wall time, GPU-busy union,
idle gaps, and what runs during the phases of a step (by stream)."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Stream_Id"]),
       re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])) for r in rows]
ev.sort()
# step boundaries: the fused Adam kernels end a step
ends = [e for s, e, st, n in ev if "FusedAdam" in n or "adam_multi_kernel" in n]
# group consecutive adam launches (7 per step)
bounds = []
for t in ends:
    if not bounds or t - bounds[-1] > 5e6:
        bounds.append(t)
    else:
        bounds[-1] = t
print("steps found:", len(bounds))
for i in range(max(1, len(bounds) - 3), len(bounds)):
    t0, t1 = bounds[i - 1], bounds[i]
    sel = [(s, e, st, n) for s, e, st, n in ev if s >= t0 and e <= t1]
    # union of busy intervals
    busy, cur_s, cur_e = 0, None, None
    for s, e, st, n in sel:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, st, n in sel)
    print("step %d: wall %.2f ms, busy(union) %.2f ms, idle %.2f ms, sum of kernels %.2f ms, launches %d"
          % (i, (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, tot / 1e6, len(sel)))
    per = collections.Counter()
    for s, e, st, n in sel: per[st] += e - s
    print("   kernel time by stream:", {k: round(v / 1e6, 2) for k, v in per.items()})
    # largest idle gaps
    gaps, last = [], t0
    cur_e = t0
    for s, e, st, n in sel:
        if s > cur_e: gaps.append((s - cur_e, n[:60]))
        cur_e = max(cur_e, e)
    gaps.sort(reverse=True)
    print("   largest gaps (us, next kernel):", [(round(g / 1e3, 1), n) for g, n in gaps[:6]])

# ---- per-stream view of the last full step -------------------------------------------------
t0, t1 = bounds[-2], bounds[-1]
sel = [(s, e, st, n) for s, e, st, n in ev if s >= t0 and e <= t1]
streams = sorted(set(st for _, _, st, _ in sel))
print("\nlast step, per stream (ms relative to step start): first start, last end, busy, #kernels")
for st in streams:
    ks = [(s, e, n) for s, e, q, n in sel if q == st]
    print("  stream %d: %.2f .. %.2f  busy %.2f  n=%d" % (st, (ks[0][0] - t0) / 1e6, (ks[-1][1] - t0) / 1e6,
                                                        sum(e - s for s, e, n in ks) / 1e6, len(ks)))
# 1 ms buckets: which streams are active
import math
nb = int(math.ceil((t1 - t0) / 1e6))
print("per-ms activity (fraction of the ms each stream has a kernel running):")
for b in range(nb):
    lo, hi = t0 + b * 1e6, t0 + (b + 1) * 1e6
    row = []
    for st in streams:
        busy = sum(max(0, min(e, hi) - max(s, lo)) for s, e, q, n in sel if q == st and e > lo and s < hi)
        row.append("%3d%%" % (100 * busy / 1e6))
    tops = [((min(e, hi) - max(s, lo)), n) for s, e, q, n in sel if e > lo and s < hi]
    top = max(tops) if tops else (0, "-")
    print("  %2d ms: %s   %s" % (b, " ".join(row), re.sub(r"<.*", "", top[1])[:40]))
