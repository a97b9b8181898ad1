#!/bin/bash
# One profiling pass of the bench step: bench line, rocprofv3 kernel stats of the same command, optional HBM traffic.
# usage: bash scripts/prof_round.sh TAG [traffic]
set -e
TAG=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout -k 10 300 python -u bench.py --kernel-table > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
tail -1 gpurun_out/$TAG/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/stats -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/$TAG/stats.log 2>&1
cd $R && python scripts/trace_summary.py $(ls gpurun_out/$TAG/stats/*/*kernel_trace.csv gpurun_out/$TAG/stats/*kernel_trace.csv 2>/dev/null | head -1) 27 > gpurun_out/$TAG/kernel_summary.txt 2>&1 || true
# the same step with the weight-gradient stream off: every kernel runs alone, so its duration is its own
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/stats_solo -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-fork > $R/gpurun_out/$TAG/stats_solo.log 2>&1
cd $R && python scripts/trace_summary.py $(ls gpurun_out/$TAG/stats_solo/*/*kernel_trace.csv gpurun_out/$TAG/stats_solo/*kernel_trace.csv 2>/dev/null | head -1) 27 > gpurun_out/$TAG/kernel_summary_solo.txt 2>&1 || true
if [ "$2" == "traffic" ]; then
  cd /tmp
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 280 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/$TAG/traffic_$C -o bench --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$TAG/traffic_$C.log 2>&1
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/$TAG/traffic_$C -o calib --output-format csv -- python3 $R/scripts/kbench.py raw_copy 2 > $R/gpurun_out/$TAG/traffic_calib_$C.log 2>&1
  done
  cd $R && python scripts/traffic_summary.py gpurun_out/$TAG > gpurun_out/$TAG/hbm_traffic.txt 2>&1 || true
fi
echo done
