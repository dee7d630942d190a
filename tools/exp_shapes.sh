#!/bin/bash
# pass time + kernel breakdown at the shapes of BASELINE configs #5 (k = 41), #4 scaled and #4 (GPU box only).
# usage: tools/exp_shapes.sh <tag> [P3_OPTS]
tag=${1:-shapes}; export P3_OPTS=${2:-}
R=$GRAFT_REPO_ROOT
export P3_PASSES=3
for shape in "10000000 41 150 5000" "200000 63 10000 10000" "1000000 63 10000 10000"; do
  n=$(echo $shape | tr ' ' '_')
  timeout -k 10 400 python3 $R/tools/exp_big.py $shape > $R/gpurun_out/${tag}_$n.log 2>&1 || exit 1
  grep -q "Memory access fault" $R/gpurun_out/${tag}_$n.log && exit 2
  echo "== $shape  [$P3_OPTS]"
  grep "^pass 2\|^k_" $R/gpurun_out/${tag}_$n.log
done
