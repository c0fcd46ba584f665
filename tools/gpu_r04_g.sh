#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04g
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_preview.py -x -q > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -12 $out/pytest.txt
timeout -k 10 300 python tools/run_c5_only.py 2048 10 sweep >> $out/c5.txt 2>&1
timeout -k 10 300 python tools/run_c5_only.py 16384 5 sweep >> $out/c5.txt 2>&1
grep -v amdgpu.ids $out/c5.txt
timeout -k 10 600 bash tools/profile_kernel.sh r04g_c5prof ltv_sweep -- python3 $R/tools/run_c5_only.py 2048 5 sweep > $out/c5prof.txt 2>&1
grep -v amdgpu $out/c5prof.txt | tail -22
