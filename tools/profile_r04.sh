#!/bin/bash
# Round 4's profile set (gpurun -- 'bash tools/profile_r04.sh'): the round's bench lines and rocprofv3
# kernel stats / HBM counters of the headline step (tools/profile_round.sh), then the kernels new in this
# round under rocprofv3 in passes of their own (tools/profile_kernel.sh): C4 on the scan form, C5 on the
# sweep kernel, f2's fused goal distances.  Summaries: tools/summarize_profile.py and the summary.txt files.
R=$GRAFT_REPO_ROOT
cd $R
# (kernel arguments in device memory, as the package sets it for a process that imports it before HIP starts:
# under rocprofv3 the runtime is up before python runs)
export HIP_FORCE_DEV_KERNARG=1
bash tools/profile_round.sh r04 || echo "profile_round failed"
bash tools/profile_kernel.sh r04_c4 toeplitz_scan -- python3 $R/tools/run_tiled_only.py 8192 3 1 all 0 > $R/gpurun_out/r04_c4.log 2>&1
bash tools/profile_kernel.sh r04_c5 ltv_sweep -- python3 $R/tools/run_c5_only.py 2048 5 sweep > $R/gpurun_out/r04_c5.log 2>&1
cd $R
python3 tools/run_c5_only.py 16384 5 sweep > $R/gpurun_out/r04_c5_16384.txt 2>&1
python3 tools/run_c5_only.py 2048 5 fill > $R/gpurun_out/r04_c5_fill.txt 2>&1
python3 tools/run_tiled_only.py 8192 3 1 all 4 > $R/gpurun_out/r04_c4_paths.txt 2>&1
python3 tools/run_tiled_only.py 8192 3 0 all 0 >> $R/gpurun_out/r04_c4_paths.txt 2>&1
python3 tools/run_tiled_only.py 8192 3 0 all 3 >> $R/gpurun_out/r04_c4_paths.txt 2>&1
bash tools/gpu_r04_shared.sh > $R/gpurun_out/r04_c4_shared.txt 2>&1
python3 tools/ablate_scan.py 4096 > $R/gpurun_out/r04_c4_ablation.txt 2>&1
tools/microbench/store_rate5 > $R/gpurun_out/r04_store_rate5.txt 2>&1
tail -3 $R/gpurun_out/r04_c4/summary.txt $R/gpurun_out/r04_c5/summary.txt
head -c 1500 $R/gpurun_out/prof_r04/bench_plain.json
