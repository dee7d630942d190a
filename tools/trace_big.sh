#!/bin/bash
# kernel trace of a long-read pass (GPU box only).  usage: tools/trace_big.sh <tag> <reads> <k> <len>
tag=${1:-big}; RD=${2:-1000000}; K=${3:-63}; LEN=${4:-10000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export P3_PASSES=3
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_kt -o run -- python3 $R/tools/exp_big.py $RD $K $LEN 10000 > $R/gpurun_out/${tag}_kt.log 2>&1 || exit 1
grep -q "Memory access fault" $R/gpurun_out/${tag}_kt.log && exit 2
python3 $R/tools/trace_dist.py $R/gpurun_out/${tag}_kt/run_kernel_trace.csv > $R/gpurun_out/${tag}_dist.txt
grep -v "^[EW]2026" $R/gpurun_out/${tag}_kt.log | tail -8
cat $R/gpurun_out/${tag}_dist.txt
rm -f $R/gpurun_out/${tag}_kt/run_kernel_trace.csv   # (hundreds of MB for this many launches)
