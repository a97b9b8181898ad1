#!/bin/bash
# rocprofv3 kernel trace of the segmentation train step (3 warm-up + 5 timed steps); output under gpurun_out/prof_seg
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seg -o out --output-format csv -- python3 $R/scripts/seg_bench.py --batch 64 --steps 5 > $R/gpurun_out/prof_seg.log 2>&1
grep segmentation $R/gpurun_out/prof_seg.log
python3 $R/scripts/trace_summary.py $R/gpurun_out/prof_seg/out_kernel_trace.csv 8 > $R/gpurun_out/prof_seg_summary.txt
