#!/bin/bash
# the shared-model form of the tiled kernel on C4 (S, U read from memory): per-kernel times
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04sh
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}   # (as the package sets it; under rocprofv3 the runtime starts before python)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/tools/run_tiled_only.py 8192 3 0 all 0 > $out/stats.log 2>&1 || exit 1
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %5s  avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r.get("Percentage", "")))
PY
tail -1 $out/stats.log
