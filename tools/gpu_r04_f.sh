#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04f
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sweep.py -x -q > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -3 $out/pytest.txt
timeout -k 10 300 python tools/run_c5_only.py 2048 10 sweep >> $out/c5.txt 2>&1
timeout -k 10 300 python tools/run_c5_only.py 16384 5 sweep >> $out/c5.txt 2>&1
grep -v amdgpu.ids $out/c5.txt
timeout -k 10 600 bash tools/profile_kernel.sh r04f_c5prof ltv_sweep -- python3 $R/tools/run_c5_only.py 2048 5 sweep > $out/c5prof.txt 2>&1
grep -v amdgpu $out/c5prof.txt | tail -22
cd $R
timeout -k 10 600 tools/microbench/store_rate5 > $out/store_rate5.txt 2>&1
cat $out/store_rate5.txt
