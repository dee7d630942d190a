#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of known-size access shapes (tools/pmc_calib.hip), separate --pmc passes (GPU box only).
# usage: bash tools/pmc_calib.sh  -> gpurun_out/pmc_calibration.json
R=$GRAFT_REPO_ROOT
[ -x $R/tools/pmc_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $R/tools/pmc_calib $R/tools/pmc_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/calib_$ctr -o run -- $R/tools/pmc_calib > $R/gpurun_out/calib_$ctr.json 2> $R/gpurun_out/calib_$ctr.err || exit 1
done
python3 $R/tools/pmc_calib.py $R/gpurun_out/calib_FETCH_SIZE.json $R/gpurun_out/calib_FETCH_SIZE $R/gpurun_out/calib_WRITE_SIZE > $R/gpurun_out/pmc_calibration.json
