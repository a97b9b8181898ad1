#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3e
timeout -k 10 300 python -u -m pytest tests/test_bf16_training_gpu.py -x -q -s -k "converges" --timeout 200 --timeout-method thread > gpurun_out/r3e/conv.log 2>&1; grep -v "^$" gpurun_out/r3e/conv.log | tail -12
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e/tests.log 2>&1 || { tail -40 gpurun_out/r3e/tests.log; exit 1; }
tail -3 gpurun_out/r3e/tests.log
