#!/bin/bash
# scripts/kpmc.sh for several builds of the library on one box: bash scripts/kpmc_multi.sh "A.so B.so" cases [iters]
LIBS=$1; CASES=$2; IT=${3:-6}
L=medical-image-analysis_amd/libmia_hip.so
cp $L /tmp/kpmc_tree.so
echo "== tree"; bash scripts/kpmc.sh kpmc_tree $CASES $IT 2>&1 | grep -v amdgpu.ids
for b in $LIBS; do
  cp $b $L
  echo "== $b"; bash scripts/kpmc.sh kpmc_$(basename $b .so) $CASES $IT 2>&1 | grep -v amdgpu.ids
done
cp /tmp/kpmc_tree.so $L
