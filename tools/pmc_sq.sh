#!/bin/bash
# SQ counter passes over the single-GPU pass (GPU box only).  usage: tools/pmc_sq.sh <tag>
tag=${1:-sq}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$i -o run -- python3 $R/tools/exp_kernels.py 31 quick > $R/gpurun_out/${tag}_$i.log 2>&1 || exit 1
done
python3 $R/tools/pmc_agg.py $R/gpurun_out/${tag}_1 $R/gpurun_out/${tag}_2 $R/gpurun_out/${tag}_3 > $R/gpurun_out/${tag}_summary.txt
python3 $R/tools/pmc_ceilings.py 130000000 4 $R/gpurun_out/${tag}_1 $R/gpurun_out/${tag}_2 $R/gpurun_out/${tag}_3 > $R/gpurun_out/${tag}_ceilings.json
