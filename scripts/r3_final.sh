#!/bin/bash
# final check of a tree: GPU tests, smoke, the bench line (with the CPU baseline)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3_final
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_final/tests.log 2>&1 || { tail -40 gpurun_out/r3_final/tests.log; exit 1; }
tail -2 gpurun_out/r3_final/tests.log
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -1 || exit 1
timeout -k 10 300 python bench.py > gpurun_out/r3_final/bench.json 2> gpurun_out/r3_final/bench.err || exit 1
python -c "
import json
d=json.loads(open('gpurun_out/r3_final/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], 'traffic', d['roofline']['traffic'], 'launches', d['config']['kernel_launches_per_step'], 'cpu', d['cpu_baseline'])"
