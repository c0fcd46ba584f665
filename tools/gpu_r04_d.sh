#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04d
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_tiled.py tests/test_gpu_dist.py -x -q > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -5 $out/pytest.txt
for g in 4 2 1 8; do
  MPCASM_SCAN_GROUP=$g timeout -k 10 300 python tools/run_tiled_only.py 8192 3 1 all 0 >> $out/c4.txt 2>&1
done
grep -v amdgpu.ids $out/c4.txt
timeout -k 10 600 bash tools/profile_kernel.sh r04d_c5prof ltv_sweep -- python3 $R/tools/run_c5_only.py 2048 5 sweep > $out/c5prof.txt 2>&1
tail -40 $out/c5prof.txt
cd $R
timeout -k 10 900 python bench.py --steps 500 > $out/bench.json 2> $out/bench.err
echo "bench rc $?"; cat $out/bench.json | head -c 2500; tail -c 600 $out/bench.err | grep -v "^{" 
