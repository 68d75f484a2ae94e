"""Join rocprofv3's HIP runtime trace with its kernel trace (tools/lead_probe.sh): for every kernel of
the last traced step, how long before it STARTED had the host finished queueing it (lead), and what the
host was doing during the large GPU gaps."""
import csv
import glob
import re
import sys

out = sys.argv[1]
kt = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
ht = glob.glob(out + "/**/*hip_api_trace.csv", recursive=True)[0]
K = list(csv.DictReader(open(kt)))
H = list(csv.DictReader(open(ht)))
print("kernels", len(K), "api calls", len(H))
print("kernel cols", list(K[0].keys()))
print("api cols", list(H[0].keys()))
api = {}
for r in H:
    api[r["Correlation_Id"]] = r
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)[:48]
K.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [short(r["Kernel_Name"]) for r in K]
adam = [i for i, n in enumerate(names) if n.startswith("adam_multi")]
a, b = adam[-2] + 1, adam[-1] + 1
seg = K[a:b]
t0 = int(seg[0]["Start_Timestamp"])
print("last step: %d kernels, %.2f ms" % (len(seg), (int(seg[-1]["End_Timestamp"]) - t0) / 1e6))
# host API activity in that window, by function
h0 = min(int(api[r["Correlation_Id"]]["Start_Timestamp"]) for r in seg if r["Correlation_Id"] in api)
h1 = max(int(api[r["Correlation_Id"]]["End_Timestamp"]) for r in seg if r["Correlation_Id"] in api)
print("host queued this step's kernels over %.2f ms (first call %.2f ms before the first kernel started)"
      % ((h1 - h0) / 1e6, (t0 - h0) / 1e6))
by = {}
for r in H:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= h0 and e <= h1:
        d = by.setdefault(r["Function"], [0, 0])
        d[0] += 1
        d[1] += e - s
for f, (n, d) in sorted(by.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-36s %6d calls %8.2f ms" % (f, n, d / 1e6))
# lead per kernel, bucketed by ms of the step
prev_end = None
rows = []
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    c = api.get(r["Correlation_Id"])
    lead = (s - int(c["End_Timestamp"])) / 1e3 if c else float("nan")
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = max(prev_end or 0, e)
    rows.append(((s - t0) / 1e6, lead, gap, short(r["Kernel_Name"]), c["Function"] if c else "?"))
print("\nper ms of the step: kernels started, median lead (us), min lead (us)")
import statistics
for ms in range(int(rows[-1][0]) + 1):
    ls = [x[1] for x in rows if int(x[0]) == ms and x[1] == x[1]]
    if ls:
        print("  %3d ms: %4d kernels  median lead %9.0f  min %9.0f" % (ms, len(ls), statistics.median(ls), min(ls)))
print("\nGPU gaps > 40 us (no kernel running on any stream): at ms, gap us, lead us of the next kernel, kernel, api")
for x in rows:
    if x[2] > 40:
        print("  %7.2f  gap %7.0f  lead %9.0f  %-48s %s" % x)
