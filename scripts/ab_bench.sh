#!/bin/bash
# Same-box A/B of two builds of the library: bash scripts/ab_bench.sh ALT.so [rounds] [bench flags...]
# Alternates the in-tree library and ALT.so under bench.py, so box-to-box and thermal drift (2-3 % between boxes,
# under 0.2 % between alternating runs on one box) cancel. Run on the GPU box only (it swaps the library in its copy).
set -e
ALT=$1; N=${2:-3}; shift; shift || true
L=medical-image-analysis_amd/libmia_hip.so
cp $L /tmp/ab_new.so
for i in $(seq $N); do
  for v in new alt; do
    if [ $v == alt ]; then cp $ALT $L; else cp /tmp/ab_new.so $L; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done
cp /tmp/ab_new.so $L
