#!/bin/bash
# bench.py's C2 variants under rocprofv3, one process each (gpurun -- 'bash tools/profile_variants.sh'):
# kernel time of the 34- and 36-wide buckets beside the hipEvent time per launch
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/variants
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 34 36; do
  rocprofv3 --kernel-trace --stats -d $out/p$v -o p --output-format csv -- python3 $R/tools/run_variant.py $v full 2000 > $out/v$v.log 2>&1 || exit 1
  grep "per launch" $out/v$v.log
  grep -h "resident" $out/p$v/*kernel_stats.csv | cut -c1-200
done
