#!/bin/bash
# Same-box comparison of several alternate builds of the library against the in-tree one:
#   bash scripts/ab_multi.sh ROUNDS ALT1.so ALT2.so ...     (GPU box only: it swaps the library in its copy)
set -e
N=$1; shift
L=medical-image-analysis_amd/libmia_hip.so
cp $L /tmp/ab_new.so
for i in $(seq $N); do
  for v in /tmp/ab_new.so "$@"; do
    cp $v $L
    timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'])"
  done
done
cp /tmp/ab_new.so $L
