#!/bin/bash
# Counter pass over scripts/kbench.py cases (kernel-trace only beside it): bash scripts/kpmc.sh TAG case1,case2 [iters]
# Prints per kernel: time, waves resident per SIMD, vector-ALU / LDS issue share of the SIMD cycles, LDS bank-conflict share.
set -e
TAG=$1; CASES=$2; IT=${3:-6}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $R/gpurun_out/$TAG/pmc -o kb --output-format csv -- python3 $R/scripts/kbench.py $CASES $IT > $R/gpurun_out/$TAG/kpmc.log 2>&1
cd $R
python - <<PY
import csv, collections, re
pm = collections.defaultdict(dict)
for r in csv.DictReader(open("gpurun_out/$TAG/pmc/kb_counter_collection.csv")):
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:60]
    pm[(int(r["Dispatch_Id"]), n, r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"])][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for (d, n, g, l, v), c in pm.items():
    agg[(n, g, l, v)].append(c)
print("kernel | grid lds vgpr | cycles | waves/SIMD | VALU share | LDS share | LDS conflict/LDS busy | VALU inst/wave | LDS inst/wave")
for (n, g, l, v), cs in agg.items():
    if len(cs) < 3: continue
    c = cs[len(cs) // 2]
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    simd_cyc = cyc * 1024
    waves = float(g) / 64
    print(f"{n} | {g} {l} {v} | {cyc:.0f} | {4 * c['SQ_WAVE_CYCLES'] / simd_cyc:.2f} | {4 * c['SQ_ACTIVE_INST_VALU'] / simd_cyc:.2f} | "
          f"{4 * c['SQ_ACTIVE_INST_LDS'] / simd_cyc:.2f} | {c['SQ_LDS_BANK_CONFLICT'] / max(4 * c['SQ_ACTIVE_INST_LDS'], 1):.2f} | "
          f"{c['SQ_INSTS_VALU'] / waves:.0f} | {c['SQ_INSTS_LDS'] / waves:.0f}")
PY
