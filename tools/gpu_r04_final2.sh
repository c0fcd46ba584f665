#!/bin/bash
# end of round 4, after the scan form's set-up was fused: the C4 profile and the bench lines once more
R=$GRAFT_REPO_ROOT
cd $R
export HIP_FORCE_DEV_KERNARG=1
bash tools/profile_kernel.sh r04_c4 toeplitz_scan -- python3 $R/tools/run_tiled_only.py 8192 3 1 all 0 > $R/gpurun_out/r04_c4.log 2>&1
cd $R
for a in "1 all 1" "1 all 4" "0 all 0" "0 all 3"; do python3 tools/run_tiled_only.py 8192 3 $a 2>&1 | grep C4; done > $R/gpurun_out/r04_c4_paths.txt
python3 tools/ablate_scan.py 4096 2>&1 | grep -v amdgpu > $R/gpurun_out/r04_c4_ablation.txt
python3 bench.py > $R/gpurun_out/prof_r04/bench_plain.json 2> $R/gpurun_out/prof_r04/bench_plain.log
head -c 1900 $R/gpurun_out/prof_r04/bench_plain.json; echo
grep "toeplitz_scan\|lti_tables\|compose_d\|HBM bytes" $R/gpurun_out/r04_c4/summary.txt
cat $R/gpurun_out/r04_c4_paths.txt
