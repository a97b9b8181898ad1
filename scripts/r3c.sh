#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3c
timeout -k 10 300 python -u -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_nt" --timeout 120 --timeout-method thread > gpurun_out/r3c/tests.log 2>&1 || { tail -40 gpurun_out/r3c/tests.log; exit 1; }
tail -2 gpurun_out/r3c/tests.log
timeout -k 10 200 python scripts/fin_bench.py 2>&1 | tee gpurun_out/r3c/fin_bench.txt
timeout -k 10 300 python -u -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py tests/test_bf16_training_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3c/tests2.log 2>&1 || { tail -40 gpurun_out/r3c/tests2.log; exit 1; }
tail -2 gpurun_out/r3c/tests2.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline 2>gpurun_out/r3c/bench.err | tee gpurun_out/r3c/bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused', d['ms_per_step'], d['config']['kernel_launches_per_step'])" || exit 1
done
