#!/bin/bash
# the persistent kernel's set-up: instruction-cache misses per launch, by batch size
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r04ic
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export MPCASM_LTI=1
for B in 512 4096 65536; do
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES -d $out/ic$B -o p --output-format csv -- python3 $R/tools/run_assemble_only.py $B > $out/ic$B.log 2>&1 || echo "ic pass failed"
done
python3 - $out <<'PY' > $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
for B in (512, 4096, 65536):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/ic%d/**/*counter_collection.csv" % B, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("B =", B)
    for k, v in sorted(acc.items()):
        print("  %-40s %-30s %s" % (k[0], k[1], " ".join("%.4g" % x for x in v)))
PY
cat $out/summary.txt
