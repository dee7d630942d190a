#!/bin/bash
# BASELINE config #5 on one GPU: 10 M x 150 bp, five independent runs at k = 21, 31, 41, 51, 61 (timing + statistics).
# usage (GPU box): bash tools/exp_config5.sh  -> gpurun_out/config5.log
R=${GRAFT_REPO_ROOT:-.}
: > $R/gpurun_out/config5.log
for k in 21 31 41 51 61; do
  echo "== k=$k" >> $R/gpurun_out/config5.log
  P3_PASSES=3 timeout -k 10 400 python3 $R/tools/exp_big.py 10000000 $k >> $R/gpurun_out/config5.log 2>&1 || exit 1
done
grep "^==\|^pass 2\|^{" $R/gpurun_out/config5.log | cut -c1-220
