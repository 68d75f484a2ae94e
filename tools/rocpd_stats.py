"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / mean duration."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name) if not name.startswith("void") else re.sub(r"\((?!.*<).*", "", name)
    a = agg.setdefault(name, [0, 0])
    a[0] += 1
    a[1] += e - s
tot = sum(v[1] for v in agg.values())
t0 = min(r[1] for r in rows); t1 = max(r[2] for r in rows)
print("kernels: %d dispatches, busy %.2f ms over %.2f ms wall" % (len(rows), tot / 1e6, (t1 - t0) / 1e6))
print("%-110s %7s %11s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-110s %7d %11.3f %10.1f %6.2f" % (name[:110], n, t / 1e6, t / n / 1e3, 100.0 * t / tot))
