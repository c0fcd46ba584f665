#!/bin/bash
# gpurun -- 'bash tools/pmc_placement.sh': counters of the last six launches of tools/pmc_placement.py
# (three into the fastest placement of C3's results, three into the slowest), one pass per counter group
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_placement
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
             "TCC_EA0_WRREQ_LEVEL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum" \
             "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCC_WRITE_SECTORS_sum" \
             "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" \
             "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group -d $out/g$i -o p --output-format csv -- python3 $R/tools/pmc_placement.py > $out/g$i.log 2>&1 || { echo "group $i failed"; tail -3 $out/g$i.log; continue; }
  grep "fastest" $out/g$i.log
  python3 - $out/g$i <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "resident" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
last = list(by.items())[-6:]
for name in sorted(last[0][1]):
    vals = [d[name] for _, d in last]
    print("  %-48s fast %s   slow %s" % (name, " ".join("%.4g" % v for v in vals[:3]), " ".join("%.4g" % v for v in vals[3:])))
PY
done
