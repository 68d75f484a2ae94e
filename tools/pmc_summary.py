"""Fold rocprofv3 --pmc counter_collection CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, os, re, sys
root = sys.argv[1]
agg = {}
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)[:90]
        d = agg.setdefault(name, {})
        c = d.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"]); c[1] += 1
for name, d in agg.items():
    if not any(k in name for k in ("conv_", "bn_", "maxpool", "gemm", "wgrad", "nce_", "positive_mask", "adam", "loss", "stage")):
        continue
    print(name)
    for k, (s, n) in sorted(d.items()):
        print("   %-32s %16.1f  (n=%d)" % (k, s / n, n))
