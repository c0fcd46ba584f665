#!/bin/bash
# round 4: sweep kernel with tables in LDS; scan kernel with prefetched tickets; set-up stamps of the persistent kernel
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04c
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_tiled.py -x -q > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -5 $out/pytest.txt
timeout -k 10 300 python tools/run_c5_only.py 2048 10 sweep >> $out/c5.txt 2>&1
timeout -k 10 300 python tools/run_c5_only.py 16384 5 sweep >> $out/c5.txt 2>&1
for g in 4 8 2; do
  MPCASM_SCAN_GROUP=$g timeout -k 10 300 python tools/run_tiled_only.py 8192 3 1 all 0 >> $out/c4.txt 2>&1
done
timeout -k 10 300 python tools/ablate_scan.py 4096 >> $out/c4.txt 2>&1
MPCASM_LTI=1 MPCASM_JIT=1 timeout -k 10 300 python tools/stamp_resident.py 4096 > $out/stamps.txt 2>&1
grep -v amdgpu.ids $out/c5.txt $out/c4.txt $out/stamps.txt
