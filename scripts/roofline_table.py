"""Per-kernel HBM roofline table of one training step from two committed profile files:
    python scripts/roofline_table.py profiles/r3f_hbm_traffic.txt profiles/r3f_alone_rocprofv3_kernel_stats.csv [steps]
bytes = HBM traffic from the counter passes (scripts/traffic_summary.py), time = the kernel's average duration with every
kernel alone on the chip (bench.py --no-fork under rocprofv3), both per step; peak = 8 TB/s."""
import csv
import re
import sys

PEAK = 8000.0   # GB/s


def short(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name)).strip()


def main():
    per_kernel = {}
    for line in open(sys.argv[1]):
        m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            per_kernel[m.group(1).strip()] = (float(m.group(3)), float(m.group(4)))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 27
    times = {}
    for r in csv.DictReader(open(sys.argv[2])):
        n = short(r["Name"])
        t = times.setdefault(n, [0, 0.0])
        t[0] += int(r["Calls"])
        t[1] += float(r["TotalDurationNs"])
    rows = []
    for n, k in per_kernel.items():
        t = times.get(short(n))
        if not t:
            continue
        ms = t[1] / steps / 1e6
        gb = k[0] + k[1]
        rows.append((ms, t[0] / steps, gb, short(n)))
    rows.sort(reverse=True)
    print("| kernel | launches/step | ms/step alone | HBM GB/step | TB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    tot_ms = tot_gb = 0.0
    for ms, n, gb, name in rows:
        tot_ms += ms
        tot_gb += gb
        if ms >= 0.05:
            print(f"| `{name}` | {n:.0f} | {ms:.3f} | {gb:.3f} | {gb / ms:.2f} | {gb / ms * 1e3 / PEAK:.2f} |")
    print(f"| all kernels with counters | | {tot_ms:.2f} | {tot_gb:.1f} | {tot_gb / tot_ms:.2f} | {tot_gb / tot_ms * 1e3 / PEAK:.2f} |")


if __name__ == "__main__":
    main()
