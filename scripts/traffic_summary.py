"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh: calibrate the counter unit on the 512 MiB
device copy, then report HBM bytes per launch for each kernel of the bench step (last step's launches only)."""
import csv, sys, re, json, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
steps_in_run = 4   # bench ran 2 warm-up + 2 timed steps

def load(path):
    rows = list(csv.DictReader(open(path)))
    return [(re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:60], float(r["Counter_Value"])) for r in rows]

unit = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = [v for k, v in load(f"{root}/traffic_{c}/calib_counter_collection.csv") if "elementwise" in k or "copy" in k.lower()]
    big = sorted(cal)[-2:]   # the copy launches (512 MiB read, 512 MiB written each)
    unit[c] = 512 * 1024 * 1024 / (sum(big) / len(big))
    print(f"{c}: copy of 512 MiB counted as {sum(big)/len(big):.4g} -> {unit[c]:.1f} bytes per count")
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for k, v in load(f"{root}/traffic_{c}/bench_counter_collection.csv"):
        per[k].append(v)
    for k, v in per.items():
        n = len(v) // steps_in_run
        if n == 0:
            continue
        last = v[-n:]
        out.setdefault(k, {})[c] = (n, sum(last) * unit[c])
tot_r = tot_w = 0.0
print(f"{'kernel':60s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s}  per step")
for k, d in sorted(out.items(), key=lambda kv: -(kv[1].get('FETCH_SIZE', (0, 0))[1] + kv[1].get('WRITE_SIZE', (0, 0))[1]))[:24]:
    n = d.get("FETCH_SIZE", d.get("WRITE_SIZE"))[0]
    r, w = d.get("FETCH_SIZE", (0, 0))[1], d.get("WRITE_SIZE", (0, 0))[1]
    print(f"{k:60s} {n:8d} {r/1e9:9.3f} {w/1e9:9.3f}")
for k, d in out.items():
    tot_r += d.get("FETCH_SIZE", (0, 0))[1]; tot_w += d.get("WRITE_SIZE", (0, 0))[1]
print(f"total per step: read {tot_r/1e9:.2f} GB, write {tot_w/1e9:.2f} GB")
fam = collections.defaultdict(lambda: [0, 0.0])
for k, d in out.items():
    name = k.split("<")[0].replace("_kernel", "")
    n = d.get("FETCH_SIZE", d.get("WRITE_SIZE"))[0]
    fam[name][0] += n
    fam[name][1] += d.get("FETCH_SIZE", (0, 0))[1] + d.get("WRITE_SIZE", (0, 0))[1]
json.dump({"unit_bytes_per_count": unit, "families": {k: {"launches_per_step": v[0], "hbm_bytes_per_step": v[1],
           "hbm_bytes_per_launch": v[1] / max(v[0], 1)} for k, v in fam.items()}},
          open(f"{root}/traffic_summary.json", "w"), indent=1)
