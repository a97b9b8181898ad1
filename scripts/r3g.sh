#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3g
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_bf16_training_gpu.py tests/test_seg_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3g/tests.log 2>&1 || { tail -40 gpurun_out/r3g/tests.log; exit 1; }
tail -2 gpurun_out/r3g/tests.log
L=medical-image-analysis_amd/libmia_hip.so
for i in 1 2 3; do
timeout -k 10 200 python bench.py --no-cpu-baseline 2>gpurun_out/r3g/bench.err | tee gpurun_out/r3g/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['config']['kernel_launches_per_step'])" || exit 1
done
