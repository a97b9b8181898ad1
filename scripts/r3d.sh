#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3d
timeout -k 10 400 python -u -m pytest tests/test_preprocess_gpu.py tests/test_model_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r3d/tests.log 2>&1 || { tail -40 gpurun_out/r3d/tests.log; exit 1; }
tail -2 gpurun_out/r3d/tests.log
timeout -k 10 200 python scripts/preprocess_bench.py > gpurun_out/r3d/preprocess_bench.jsonl 2>gpurun_out/r3d/preprocess.err; cut -c1-260 gpurun_out/r3d/preprocess_bench.jsonl | head -12
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline 2>gpurun_out/r3d/bench.err | tee gpurun_out/r3d/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nhwc-bf16 input', d['ms_per_step'], d['config']['kernel_launches_per_step'])" || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --nchw-input 2>gpurun_out/r3d/bench2.err | tee gpurun_out/r3d/bench_nchw.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nchw-f32 input', d['ms_per_step'], d['config']['kernel_launches_per_step'])" || exit 1
done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3d/pp -o pp --output-format csv -- python3 $R/scripts/preprocess_bench.py > $R/gpurun_out/r3d/pp.log 2>&1; cd $R; python - <<'PY'
import csv,glob,collections,re
f=glob.glob('gpurun_out/r3d/pp/**/*kernel_stats.csv',recursive=True)
for p in f:
    for r in csv.DictReader(open(p)):
        n=r['Name']
        if 'slice' in n or 'resize' in n or 'gather' in n:
            print(n[:70], r['Calls'], 'avg_ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
