#!/bin/bash
# HBM bytes of the shared-model form's two streaming kernels (C4, one system for 8192 instances)
R=$GRAFT_REPO_ROOT
cd $R
export HIP_FORCE_DEV_KERNARG=1
bash tools/profile_kernel.sh r04_shg shared_g -- python3 $R/tools/run_tiled_only.py 8192 3 0 all 0 > $R/gpurun_out/r04_shg.log 2>&1
cd $R
bash tools/profile_kernel.sh r04_shp shared_p -- python3 $R/tools/run_tiled_only.py 8192 3 0 all 0 > $R/gpurun_out/r04_shp.log 2>&1
cd $R
grep "shared_\|HBM bytes\|WRITE_SIZE\|FETCH_SIZE" gpurun_out/r04_shg/summary.txt gpurun_out/r04_shp/summary.txt
