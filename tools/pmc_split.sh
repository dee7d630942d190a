#!/bin/bash
# SQ counters of the owner-side k_split under the two pre-aggregation layouts (GPU box only): AGG_LAYOUT=1 blocked, 3 = mixed
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lay in 1 3; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
             "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT"; do
    i=$((i+1))
    AGG_LAYOUT=$lay SHARD=minimizer rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/split_${lay}_$i -o run -- python3 $R/tools/exp_loopback_scale.py 8 1000000 > $R/gpurun_out/split_${lay}_$i.log 2>&1 || exit 1
  done
  python3 $R/tools/pmc_agg.py $R/gpurun_out/split_${lay}_1 $R/gpurun_out/split_${lay}_2 $R/gpurun_out/split_${lay}_3 > $R/gpurun_out/split_${lay}_summary.txt
done
