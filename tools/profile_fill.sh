#!/bin/bash
# K1 alone under rocprofv3 on the GPU box (gpurun -- 'bash tools/profile_fill.sh r02'): per
# north-star shape one --kernel-trace --stats pass and one --pmc WRITE_SIZE pass (counters
# in a pass of their own); tools/summarize_fill_profile.py condenses them into profiles/.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/fill_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "3 1 16 0 4096" "3 1 16 0 8192" "3 1 16 0 65536" "12 6 64 0 1024" "3 1 100 1 2048" "3 1 100 1 16384"; do
  i=$((i+1))
  echo "$c" > $out/case$i.txt
  rocprofv3 --kernel-trace --stats -d $out/stats$i -o p --output-format csv -- python3 $R/tools/run_fill_only.py $c 30 > $out/stats$i.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write$i -o p --output-format csv -- python3 $R/tools/run_fill_only.py $c 10 > $out/write$i.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch$i -o p --output-format csv -- python3 $R/tools/run_fill_only.py $c 10 > $out/fetch$i.log 2>&1 || exit 1
done
cd $R && python3 tools/summarize_fill_profile.py $out $tag
