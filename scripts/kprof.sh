#!/bin/bash
# rocprofv3 kernel durations of scripts/kbench.py cases: bash scripts/kprof.sh TAG case1,case2 [iters]
set -e
TAG=$1; CASES=$2; IT=${3:-20}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/kp -o kb --output-format csv -- python3 $R/scripts/kbench.py $CASES $IT > $R/gpurun_out/$TAG/kbench.log 2>&1
cd $R
python - <<PY
import csv, collections, re
rows = list(csv.DictReader(open("gpurun_out/$TAG/kp/kb_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))[:70]
    d[(n, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 5:
        v2 = sorted(v)
        print(f"{len(v):5d} x  med {v2[len(v2)//2]:9.1f} us  min {v2[0]:9.1f}  grid {k[1]:>7}x{k[2]:>4}x{k[3]:>4}  {k[0]}")
PY
grep "us " gpurun_out/$TAG/kbench.log || true
