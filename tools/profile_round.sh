#!/bin/bash
# rocprofv3 evidence for one build (GPU box only): kernel-trace stats of the bench command, then FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes.  usage: tools/profile_round.sh <tag>; results in gpurun_out/<tag>_*
tag=${1:-cur}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scale-base > $R/gpurun_out/${tag}_stats_bench.json 2> $R/gpurun_out/${tag}_stats.err || exit 1
cp $R/gpurun_out/${tag}_stats/run_kernel_stats.csv $R/gpurun_out/${tag}_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_pmc_$ctr -o run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scale-base --cold-steps 0 > $R/gpurun_out/${tag}_pmc_$ctr.json 2> $R/gpurun_out/${tag}_pmc_$ctr.err || exit 1
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE $R/gpurun_out/${tag}_pmc_WRITE_SIZE > $R/gpurun_out/${tag}_pmc_traffic.json
