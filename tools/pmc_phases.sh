#!/bin/bash
# gpurun -- 'bash tools/pmc_phases.sh': per-phase instruction counts of the persistent kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_ph
rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS} -d $R/gpurun_out/pmc_ph -o p --output-format csv -- python3 $R/tools/pmc_phases.py 4096 > $R/gpurun_out/pmc_ph.log 2>&1 || exit 1
cd $R && python3 - <<PY
import csv, glob, sys
sys.path.insert(0, "tools")
rows = {}
for f in glob.glob("gpurun_out/pmc_ph/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "resident_" in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = ["all", "none", "staging", "compose", "hessian", "constraints", "Pq-store"]
ids = sorted(rows)
cols = sorted({c for d in rows.values() for c in d})
print("%-12s" % "phases on" + "".join("%14s" % c.replace("SQ_INSTS", "I").replace("SQ_", "")[:13] for c in cols) + "   (per instance, B=4096)")
for k, name in enumerate(names):
    d = rows[ids[2 * k + 1]]
    print("%-12s" % name + "".join("%14.1f" % (d.get(c, 0) / 4096) for c in cols))
PY
