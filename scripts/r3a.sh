#!/bin/bash
# round-3 first pass: GPU tests, then same-box A/B of the gemm_tn workgroup map, then the graphed step
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3a
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3a/tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3a/tests.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; exit $rc; fi
for i in 1 2; do
  for v in 1 0; do
    MIA_TN_XCD=$v timeout -k 10 200 python bench.py --no-cpu-baseline 2>gpurun_out/r3a/bench_xcd$v.err | tee gpurun_out/r3a/bench_xcd$v.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd$v', d['ms_per_step'], d['roofline']['families_alone']['gemm_tn'], d['config']['kernel_launches_per_step'])" || exit 1
  done
done
timeout -k 10 300 python bench.py --no-cpu-baseline --graph 2>gpurun_out/r3a/bench_graph.err | tee gpurun_out/r3a/bench_graph.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph', d['ms_per_step'])" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --graph --no-fork 2>gpurun_out/r3a/bench_graph_nofork.err | tee gpurun_out/r3a/bench_graph_nofork.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph nofork', d['ms_per_step'])" || exit 1
timeout -k 10 200 python scripts/preprocess_bench.py > gpurun_out/r3a/preprocess_bench.jsonl 2>gpurun_out/r3a/preprocess.err; tail -12 gpurun_out/r3a/preprocess_bench.jsonl | cut -c1-300
