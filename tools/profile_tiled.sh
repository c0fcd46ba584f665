#!/bin/bash
# C4 on the tiled kernel under rocprofv3 (gpurun -- 'bash tools/profile_tiled.sh r03 [batch]'): kernel
# stats at the per-GPU batch in one call, then the matrix-core counters and the HBM traffic counters
# in passes of their own; tools/summarize_tiled_profile.py condenses them into profiles/.
tag=${1:-r03}
B=${2:-8192}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/c4_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/run_tiled_only.py $B 3 > $out/plain.txt 2>&1 || exit 1
python3 $R/tools/run_tiled_only.py $B 3 0 >> $out/plain.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/tools/run_tiled_only.py $B 3 > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS -d $out/mfma -o p --output-format csv -- python3 $R/tools/run_tiled_only.py $B 1 > $out/mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- python3 $R/tools/run_tiled_only.py $B 1 > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- python3 $R/tools/run_tiled_only.py $B 1 > $out/write.log 2>&1 || exit 1
cd $R && python3 tools/summarize_tiled_profile.py $out $tag $B
