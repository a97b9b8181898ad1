#!/bin/bash
# The round's evidence in one call: bench line, rocprofv3 kernel stats of the same command, HBM traffic passes.
set -e
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python -u bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
tail -1 gpurun_out/final_bench.json | cut -c1-200
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final_stats -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/final_stats.log 2>&1
cd $R && bash scripts/pmc_traffic.sh
