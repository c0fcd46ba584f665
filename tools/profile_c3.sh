#!/bin/bash
# C3 (3-D LIPM N=32, B=16384, horizon matrices built on chip) under rocprofv3
# (gpurun -- 'bash tools/profile_c3.sh r02'): kernel stats, then FETCH_SIZE / WRITE_SIZE in
# passes of their own; summary into profiles/<tag>_c3_profile.txt.
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/c3_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/run_c3_only.py 16384 5 lti > $out/plain_lti.txt 2>&1 || exit 1
python3 $R/tools/run_c3_only.py 16384 5 > $out/plain_staged.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 $R/tools/run_c3_only.py 16384 10 lti > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- python3 $R/tools/run_c3_only.py 16384 3 lti > $out/fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- python3 $R/tools/run_c3_only.py 16384 3 lti > $out/write.log 2>&1 || exit 1
cd $R && python3 - <<PY
import csv, glob, os
src = "$out"
def rows(sub, pat):
    for f in glob.glob(os.path.join(src, sub, "**", pat), recursive=True):
        yield from csv.DictReader(open(f))
stat = [r for r in rows("stats", "*_kernel_stats.csv") if "resident_" in r["Name"]][0]
med = lambda v: sorted(v)[len(v) // 2]
fetch = med([float(r["Counter_Value"]) for r in rows("fetch", "*_counter_collection.csv") if "resident_" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"])
write = med([float(r["Counter_Value"]) for r in rows("write", "*_counter_collection.csv") if "resident_" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE"])
B, no, nc, ng, nparams = 16384, 96, 196, 9, 0
alg = B * 8 * (no * no + no + nc * no + nc)           # written; reads: given + (A, B) + params, < 1 %
import sys
sys.path.insert(0, "tools")
import proftrace
avg = proftrace.steady_ns(os.path.join(src, "stats"), "resident_", 10)   # the run's last 10 launches: device warm
hbm = (write + 2 * fetch) * 1024
text = "\n".join([
    open(os.path.join(src, "plain_lti.txt")).read().strip().splitlines()[-1],
    open(os.path.join(src, "plain_staged.txt")).read().strip().splitlines()[-1] + "   <- staged pipeline (S, U read from HBM)",
    "",
    "kernel %s: %s launches, average of all %.1f us (rocprofv3 --kernel-trace --stats), of the last 10 (device warm) %.1f us" % (stat["Name"].split("(")[0][-40:], stat["Calls"], float(stat["AverageNs"]) / 1e3, avg / 1e3),
    "algorithmic bytes per launch (P, q, G, h written): %.1f MB -> %.0f GB/s = %.3f of 8 TB/s" % (alg / 1e6, alg / avg, alg / avg / 8000),
    "HBM bytes per launch (WRITE_SIZE %.0f KiB + 2 x FETCH_SIZE %.0f KiB, own passes): %.1f MB = %.3f x algorithmic" % (write, fetch, hbm / 1e6, hbm / alg),
]) + "\n"
open("profiles/${tag}_c3_profile.txt", "w").write(text)
print(text)
PY
