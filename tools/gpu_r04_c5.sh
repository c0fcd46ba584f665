#!/bin/bash
# the C5 profile with the round's last sweep kernel
R=$GRAFT_REPO_ROOT
cd $R
export HIP_FORCE_DEV_KERNARG=1
bash tools/profile_kernel.sh r04_c5 ltv_sweep -- python3 $R/tools/run_c5_only.py 2048 5 sweep > $R/gpurun_out/r04_c5.log 2>&1
cd $R
python3 tools/run_c5_only.py 2048 20 sweep 2>&1 | grep C5 > $R/gpurun_out/r04_c5_16384.txt
python3 tools/run_c5_only.py 16384 5 sweep 2>&1 | grep C5 >> $R/gpurun_out/r04_c5_16384.txt
for w in cost constraints; do MPCASM_C5_WHAT=$w python3 tools/run_c5_only.py 2048 20 sweep 2>&1 | grep C5 | sed "s/^/$w alone: /"; done > $R/gpurun_out/r04_c5_parts.txt
python3 tools/run_c5_only.py 2048 5 fill 2>&1 | grep C5 > $R/gpurun_out/r04_c5_fill.txt
grep "ltv_sweep\|INSTS_VALU\|INSTS_SALU\|HBM bytes" $R/gpurun_out/r04_c5/summary.txt; cat $R/gpurun_out/r04_c5_16384.txt $R/gpurun_out/r04_c5_parts.txt
