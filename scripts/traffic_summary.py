"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_traffic.sh: calibrate the counter unit on the 512 MiB
device copy, then report HBM bytes per launch for each kernel of the bench step (last step's launches only)."""
import csv, sys, re, json, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"

def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))[:60]


def load(path):
    rows = list(csv.DictReader(open(path)))
    return [(short(r["Kernel_Name"]), float(r["Counter_Value"])) for r in rows]


def load_last_step(path):
    """(kernel, value) of the launches of the LAST complete training step: the dispatches between the last two groups
    of AdamW launches (set-up work such as the host-to-device copies of the parameters is not part of any step)."""
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    idx = [i for i, n in enumerate(names) if "adamw" in n]
    ends = [i for k, i in enumerate(idx) if k + 1 == len(idx) or idx[k + 1] - i > 50]
    s0, s1 = ends[-2] + 1, ends[-1] + 1
    return [(n, float(r["Counter_Value"])) for n, r in zip(names[s0:s1], rows[s0:s1])]


unit = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = [v for k, v in load(f"{root}/traffic_{c}/calib_counter_collection.csv") if "elementwise" in k or "copy" in k.lower()]
    big = sorted(cal)[-2:]   # the copy launches (512 MiB read, 512 MiB written each)
    unit[c] = 512 * 1024 * 1024 / (sum(big) / len(big))
    print(f"{c}: copy of 512 MiB counted as {sum(big)/len(big):.4g} -> {unit[c]:.1f} bytes per count")
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for k, v in load_last_step(f"{root}/traffic_{c}/bench_counter_collection.csv"):
        per[k].append(v)
    for k, v in per.items():
        out.setdefault(k, {})[c] = (len(v), sum(v) * unit[c])
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
import importlib.util, os
_spec = importlib.util.spec_from_file_location("mia_build", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                                                       "medical-image-analysis_amd", "build.py"))
_build = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_build)        # a plain module: no torch, no GPU
kernel_tree_hash = _build.kernel_tree_hash
json.dump({"tree": kernel_tree_hash(), "unit_bytes_per_count": unit, "families": {k: {"launches_per_step": v[0], "hbm_bytes_per_step": v[1],
           "hbm_bytes_per_launch": v[1] / max(v[0], 1)} for k, v in fam.items()}},
          open(f"{root}/traffic_summary.json", "w"), indent=1)
