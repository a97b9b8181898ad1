#!/bin/bash
# HBM traffic of one bench step, per kernel: two counter passes (FETCH_SIZE, WRITE_SIZE cannot share a pass), plus the
# same two passes over a 512 MiB device copy as a calibration of the counters' units.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 280 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/traffic_$C -o bench --output-format csv -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $R/gpurun_out/traffic_$C.log 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/traffic_$C -o calib --output-format csv -- python3 $R/scripts/kbench.py raw_copy 2 > $R/gpurun_out/traffic_calib_$C.log 2>&1
done
echo done
