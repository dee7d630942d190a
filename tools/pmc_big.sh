#!/bin/bash
# Kernel trace + SQ counter passes over a long-read pass (config #4 shape, scaled).  GPU box only.
# usage: tools/pmc_big.sh <tag> <reads> <k> <len>
tag=${1:-big}; RD=${2:-200000}; K=${3:-63}; LEN=${4:-10000}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export P3_PASSES=3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt -o run -- python3 $R/tools/exp_big.py $RD $K $LEN 10000 > $R/gpurun_out/${tag}_kt.log 2>&1 || exit 1
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$i -o run -- python3 $R/tools/exp_big.py $RD $K $LEN 10000 > $R/gpurun_out/${tag}_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_agg.py $R/gpurun_out/${tag}_1 $R/gpurun_out/${tag}_2 $R/gpurun_out/${tag}_3 > $R/gpurun_out/${tag}_summary.txt
