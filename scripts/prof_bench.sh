#!/bin/bash
# rocprofv3 kernel trace of a short bench run; output under gpurun_out/prof_<tag>
set -e
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o out --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1
tail -1 $R/gpurun_out/prof_$TAG.log
