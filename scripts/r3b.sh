#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3b
timeout -k 10 300 python -u -m pytest tests/test_preprocess_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3b/tests.log 2>&1 || { tail -30 gpurun_out/r3b/tests.log; exit 1; }
tail -2 gpurun_out/r3b/tests.log
timeout -k 10 200 python scripts/preprocess_bench.py 2>gpurun_out/r3b/preprocess.err | head -4 | cut -c1-200
timeout -k 10 200 python scripts/block_times.py 128 --no-fork > gpurun_out/r3b/block_times_nofork.txt 2>&1 && cat gpurun_out/r3b/block_times_nofork.txt
timeout -k 10 200 python scripts/block_times.py 128 > gpurun_out/r3b/block_times.txt 2>&1 && cat gpurun_out/r3b/block_times.txt
