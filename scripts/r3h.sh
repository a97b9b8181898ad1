#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3h
for i in 1 2 3; do
for v in "" 1; do
MIA_OLD_SPLITS=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>gpurun_out/r3h/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('old_splits=$v', d['ms_per_step'], d['config']['kernel_launches_per_step'])" || exit 1
done
done
