"""Per-(kernel, grid) time inside ONE training step of a rocprofv3 kernel trace of bench.py (the last complete step):
python scripts/step_shapes.py <kernel_trace.csv> [substring ...]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"])) for r in rows]
idx = [i for i, n in enumerate(names) if "adamw" in n]
ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 50]
s0, s1 = ends[-2] + 1, ends[-1] + 1
agg = collections.OrderedDict()
busy = 0.0
for r, n in zip(rows[s0:s1], names[s0:s1]):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    if not pats or any(p in n for p in pats):
        agg.setdefault((n[:58], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), []).append(d)
tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tot += sum(v)
    print(f"{sum(v):8.1f} us  n={len(v):2d} avg {sum(v) / len(v):7.1f}  {k[0]} grid {k[1]}x{k[2]}x{k[3]}")
print(f"selected {tot / 1e3:.3f} ms of {busy / 1e3:.3f} ms busy in the step, {s1 - s0} launches")
