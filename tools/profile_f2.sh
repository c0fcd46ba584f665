#!/bin/bash
# f2 under rocprofv3 (gpurun -- 'bash tools/profile_f2.sh'): kernel time of preview_staged_kernel at
# 65 536 instances and its LDS counters in a pass of their own
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/f2
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/tools/bench_f2.py 65536 > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS -d $out/lds -o p --output-format csv -- python3 $R/tools/bench_f2.py 65536 > $out/lds.log 2>&1 || exit 1
grep "^f2" $out/stats.log
python3 - $out <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "preview" in r["Name"] or "lti_tables" in r["Name"]:
            print("%-28s calls %5s  avg %9.1f us  min %9.1f us" % (re.search(r"(\w+_kernel)", r["Name"]).group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "preview" in r["Kernel_Name"]:
            acc[re.search(r"(\w+_kernel)", r["Kernel_Name"]).group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kname, cs in acc.items():
    print(kname + " (65 536 instances per launch):")
    for k, v in sorted(cs.items()):
        v.sort()
        print("  %-24s median per launch %.4g (%d launches)" % (k, v[len(v) // 2], len(v)))
PY
