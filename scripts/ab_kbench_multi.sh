#!/bin/bash
# Same-box comparison of single kernels across several builds of the library:
#   bash scripts/ab_kbench_multi.sh "A.so B.so ..." case1,case2,... [iters]     (GPU box only: swaps the library in its copy)
# The in-tree library is measured first as "tree".
set -e
LIBS=$1; CASES=$2; IT=${3:-30}
L=medical-image-analysis_amd/libmia_hip.so
cp $L /tmp/ab_tree.so
timeout -k 10 200 python scripts/kbench.py $CASES $IT 2>/dev/null | awk '{print "tree", $0}'
for b in $LIBS; do
  cp $b $L
  timeout -k 10 200 python scripts/kbench.py $CASES $IT 2>/dev/null | awk -v v=$(basename $b .so) '{print v, $0}'
done
cp /tmp/ab_tree.so $L
