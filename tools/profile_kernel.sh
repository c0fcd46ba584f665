#!/bin/bash
# One kernel under rocprofv3 (gpurun -- 'bash tools/profile_kernel.sh TAG KERNEL_SUBSTRING -- python3 tools/x.py args'):
# kernel time, then SQ counters, LDS counters and the HBM byte counters in passes of their own (never with
# a trace domain other than --kernel-trace); a summary goes to gpurun_out/TAG/summary.txt
tag=$1; pat=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$tag
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}   # (as the package sets it; under rocprofv3 the runtime starts before python)
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- "$@" > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY -d $out/sq -o p --output-format csv -- "$@" > $out/sq.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $out/lds -o p --output-format csv -- "$@" > $out/lds.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- "$@" > $out/write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- "$@" > $out/fetch.log 2>&1 || exit 1
python3 - $out "$pat" > $out/summary.txt <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
print(open(out + "/stats.log").read().strip().splitlines()[-1])
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s  avg %10.1f us  min %10.1f us  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, r.get("Percentage", "")))
acc = collections.defaultdict(list)
for d in ("sq", "lds", "write", "fetch"):
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("counters of kernels matching %r (median launch, summed over the chip):" % pat)
for k, v in sorted(acc.items()):
    v.sort()
    print("  %-26s %.5g (%d launches)" % (k, v[len(v) // 2], len(v)))
if "WRITE_SIZE" in acc and "FETCH_SIZE" in acc:
    w, f = sorted(acc["WRITE_SIZE"]), sorted(acc["FETCH_SIZE"])
    print("HBM bytes per launch (WRITE_SIZE + 2 FETCH_SIZE, KiB; MI355X_MICROARCH.md): %.1f MB" % ((w[len(w) // 2] + 2 * f[len(f) // 2]) * 1024 / 1e6))
PY
cat $out/summary.txt
