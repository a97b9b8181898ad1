"""Per-shape table of the pointwise-convolution GEMMs from a kernel trace and a counter pass over scripts/gemm_shapes.py:
time, TFLOP/s, GB/s, the bound min(MFMA peak, arithmetic intensity x HBM peak), the fraction of that bound reached and
the matrix-core busy share: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs; equals 16 cycles x the number
of 16x16x32 bf16 MFMAs of the product) over 1024 x the elapsed cycles, taken as GRBM_GUI_ACTIVE / 8 (that counter is
summed over the 8 XCDs).
usage: python scripts/gemm_table.py <cases.log> <kernel_trace.csv> <counter_collection.csv> [reps]"""
import collections
import csv
import sys

MFMA_PEAK, HBM_PEAK, SIMDS, XCDS = 2.5e15, 8.0e12, 1024, 8
cases_log, trace_csv, pmc_csv = sys.argv[1:4]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
cases = [ln.split() for ln in open(cases_log) if ln.split() and ln.split()[-1] in ("measured",) or "same_as" in ln]
measured = [c for c in cases if c[-1] == "measured"]
count = collections.Counter()
for c in cases:
    count[c[0] if c[-1] == "measured" else c[-1]] += 1

rows = [r for r in csv.DictReader(open(trace_csv)) if "gemm_nt_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
assert len(rows) == reps * len(measured), (len(rows), reps, len(measured))
pmc = collections.defaultdict(dict)
for r in csv.DictReader(open(pmc_csv)):
    if "gemm_nt_kernel" in r["Kernel_Name"]:
        pmc[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
pmc_rows = [pmc[k] for k in sorted(pmc)]
assert len(pmc_rows) == reps * len(measured), (len(pmc_rows), reps * len(measured))

print("| layer (x blocks) | M | N | K | prologue | us | TFLOP/s | GB/s | bound TFLOP/s | of bound | MFMA busy |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
tot_us = tot_bound_us = 0.0
for i, c in enumerate(measured):
    name, M, N, K, pro, stats, add = c[0], int(c[1]), int(c[2]), int(c[3]), int(c[4]), int(c[5]), int(c[6])
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[i * reps + 1:(i + 1) * reps])
    us = d[len(d) // 2]
    flops = 2.0 * M * N * K
    nbytes = (M * K + N * K + M * N * (2 if add else 1)) * 2.0
    bound = min(MFMA_PEAK, flops / nbytes * HBM_PEAK)
    p = pmc_rows[i * reps + 1:(i + 1) * reps]
    busy = sorted(q.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * max(q.get("GRBM_GUI_ACTIVE", 1.0), 1.0) / XCDS) for q in p)
    n = count[name]
    tot_us += n * us
    tot_bound_us += n * flops / bound * 1e6
    print(f"| {name} (x{n}) | {M} | {N} | {K} | {'BN+SiLU+gate' if pro else '-'} | {us:.1f} | {flops / us / 1e6:.0f} | "
          f"{nbytes / us / 1e3:.0f} | {bound / 1e12:.0f} | {flops / us / 1e6 / (bound / 1e12):.2f} | {busy[len(busy) // 2]:.3f} |")
print(f"\nall {sum(count.values())} launches of a step: {tot_us / 1e3:.2f} ms measured alone, {tot_bound_us / 1e3:.2f} ms at the "
      f"per-shape bound ({tot_bound_us / tot_us:.2f} of it reached)")
