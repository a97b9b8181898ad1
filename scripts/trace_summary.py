"""Summarise a rocprofv3 kernel trace of bench.py: per-kernel totals per step and the slowest launches."""
import csv, sys, collections, re
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:70]
tot = collections.defaultdict(float); cnt = collections.Counter()
per = collections.defaultdict(list)
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name'])
    tot[k] += d; cnt[k] += 1
    per[(k, r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z'], r.get('LDS_Block_Size', ''))].append(d)
T = sum(tot.values())
print(f"total kernel time/step {T/steps/1e3:.2f} ms over {steps} steps (all launches incl. warm-up counted)")
for k, v in sorted(tot.items(), key=lambda x: -x[1])[:32]:
    print(f"{v/steps/1e3:8.3f} ms {cnt[k]/steps:7.1f} x  {k}")
if len(sys.argv) > 3:
    pat = sys.argv[3]
    print("--- per (kernel, grid, lds) for", pat)
    items = [(k, v) for k, v in per.items() if pat in k[0]]
    for k, v in sorted(items, key=lambda x: -sum(x[1]))[:40]:
        print(f"{sum(v)/steps:9.1f} us/step  n={len(v)/steps:5.1f} avg={sum(v)/len(v):8.1f}  grid={k[1]:>14} lds={k[2]:>6} {k[0][:60]}")
