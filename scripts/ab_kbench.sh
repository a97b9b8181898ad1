#!/bin/bash
# Same-box A/B of single kernels between the in-tree library and a baseline build:
#   bash scripts/ab_kbench.sh BASE.so case1,case2,... [iters] [rounds]     (GPU box only: swaps the library in its copy)
set -e
BASE=$1; CASES=$2; IT=${3:-30}; N=${4:-2}
L=medical-image-analysis_amd/libmia_hip.so
cp $L /tmp/ab_new.so
for i in $(seq $N); do
  for v in new base; do
    if [ $v == base ]; then cp $BASE $L; else cp /tmp/ab_new.so $L; fi
    timeout -k 10 200 python scripts/kbench.py $CASES $IT 2>/dev/null | awk -v v=$v '{print v, $0}'
  done
done
cp /tmp/ab_new.so $L
