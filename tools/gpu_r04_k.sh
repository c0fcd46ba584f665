#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r04k
mkdir -p $out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_assemble.py tests/test_gpu_workspace.py tests/test_walkers.py -x -q -m gpu > $out/pytest.txt 2>&1
echo "pytest rc $?" >> $out/pytest.txt
tail -3 $out/pytest.txt
MPCASM_LTI=1 MPCASM_JIT=1 timeout -k 10 300 python tools/stamp_resident.py 4096 > $out/stamps.txt 2>&1
grep -E "set-up|wave 0:|wave 3:|wave 7:" $out/stamps.txt
timeout -k 10 600 python bench.py --steps 2000 --no-extras --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 -c "
import json; r = json.load(open('$out/bench.json')); print('value %.4g  ms/step %.5f  roofline frac %.4f  avg launch %.5f ms' % (r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['avg_launch_ms']))"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/bench.py --steps 2000 --streams 1 --no-extras --no-cpu-baseline > $out/stats.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resident" in r["Name"]:
            print("%-50s calls %6s avg %8.3f us min %8.3f us" % (r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
