#!/bin/bash
# C4 under rocprofv3 (gpurun -- 'bash tools/profile_c4.sh r02'): kernel stats of the staged
# pipeline at the per-GPU batch (8 chunks of 1024), then the matrix-core counters in a pass of
# their own; tools/summarize_c4_profile.py condenses them into profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/c4_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/run_c4_only.py 1024 8 3 > $out/plain.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/tools/run_c4_only.py 1024 8 2 > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES -d $out/mfma -o p --output-format csv -- python3 $R/tools/run_c4_only.py 1024 2 1 > $out/mfma.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- python3 $R/tools/run_c4_only.py 1024 2 1 > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- python3 $R/tools/run_c4_only.py 1024 2 1 > $out/write.log 2>&1 || exit 1
cd $R && python3 tools/summarize_c4_profile.py $out $tag
