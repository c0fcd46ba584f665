#!/bin/bash
# round 4, first GPU call: the scan form of the tiled kernel -- parity, then timings
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04a
mkdir -p $out
cd $R
tools/microbench/dpp_wave_shift > $out/dpp.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_tiled.py -x -q > $out/pytest_tiled.txt 2>&1
echo "pytest rc $?" >> $out/pytest_tiled.txt
tail -5 $out/pytest_tiled.txt
for path in 0 4; do
  timeout -k 10 300 python tools/run_tiled_only.py 8192 3 1 all $path >> $out/c4.txt 2>&1
done
timeout -k 10 300 python tools/run_tiled_only.py 8192 3 0 all 0 >> $out/c4.txt 2>&1
timeout -k 10 300 python tools/run_tiled_only.py 1024 5 1 all 0 >> $out/c4.txt 2>&1
timeout -k 10 300 python tools/ablate_scan.py 4096 >> $out/c4.txt 2>&1
cat $out/c4.txt $out/dpp.txt | grep -v amdgpu.ids
