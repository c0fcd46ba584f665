#!/usr/bin/env python3
"""Condense tools/profile_fill.sh's rocprofv3 directories into profiles/<tag>_fill_profile.{txt,json}:
per K1 case the kernel's average duration (rocprofv3 --kernel-trace --stats), the algorithmic
bytes, the fraction of the 8 TB/s roofline they give, and the HBM bytes the
counters saw (WRITE_SIZE / FETCH_SIZE in KiB, own passes; FETCH doubled on gfx950 as
MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import json
import os
import re
import sys

import proftrace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fill_rows(path, pattern, name_col):
    for f in glob.glob(os.path.join(path, "**", pattern), recursive=True):
        for row in csv.DictReader(open(f)):
            if "fill_l" in row[name_col]:
                yield row


def main():
    src, tag = sys.argv[1], sys.argv[2]
    cases = []
    i = 0
    while os.path.exists(os.path.join(src, "case%d.txt" % (i + 1))):
        i += 1
        n, m, N, ltv, batch = (int(x) for x in open(os.path.join(src, "case%d.txt" % i)).read().split())
        rec = {"n": n, "m": m, "N": N, "ltv": bool(ltv), "systems": batch}
        nbytes = (8 * (N * n * n + m * N * N * n) + 8 * (n * n + n * m) * (N if ltv else 1)) * batch
        rec["algorithmic_bytes_per_launch"] = nbytes
        for row in fill_rows(os.path.join(src, "stats%d" % i), "*_kernel_stats.csv", "Name"):
            rec["kernel"] = re.search(r"fill_l\w+(<[^>]*>)?", row["Name"]).group(0)
            rec["calls"] = int(row["Calls"])
            rec["average_ns"] = float(row["AverageNs"])
            rec["min_ns"], rec["max_ns"] = float(row["MinNs"]), float(row["MaxNs"])
        for counter, key in (("WRITE_SIZE", "write"), ("FETCH_SIZE", "fetch")):
            vals = [float(r["Counter_Value"]) for r in
                    fill_rows(os.path.join(src, "%s%d" % (key, i)), "*_counter_collection.csv", "Kernel_Name")
                    if r["Counter_Name"] == counter]
            if vals:
                rec[counter + "_KiB_median"] = sorted(vals)[len(vals) // 2]
        # the last 30 launches of the stats run: after the 40 ms the run spends warming the device
        steady = proftrace.steady_ns(os.path.join(src, "stats%d" % i), "fill_l", 30)
        if steady:
            rec["average_all_launches_ns"] = rec.get("average_ns")
            rec["average_ns"] = steady
            rec["average_of"] = "the last 30 launches of the run (rocprofv3 --kernel-trace), device warm"
        if "average_ns" in rec:
            rec["achieved_GBps"] = nbytes / rec["average_ns"]
            rec["frac_of_8TBps"] = rec["achieved_GBps"] / 8000.0
        if "WRITE_SIZE_KiB_median" in rec:
            rec["hbm_bytes_per_launch"] = 1024 * (rec["WRITE_SIZE_KiB_median"] +
                                                  2 * rec.get("FETCH_SIZE_KiB_median", 0.0))
            rec["traffic_over_algorithmic"] = rec["hbm_bytes_per_launch"] / nbytes
        cases.append(rec)
    prof = os.path.join(ROOT, "profiles")
    json.dump({"tag": tag, "command": "rocprofv3 --kernel-trace --stats | --pmc WRITE_SIZE | --pmc FETCH_SIZE "
               "-- python3 tools/run_fill_only.py n m N ltv systems", "cases": cases},
              open(os.path.join(prof, tag + "_fill_profile.json"), "w"), indent=1)
    with open(os.path.join(prof, tag + "_fill_profile.txt"), "w") as f:
        f.write("%-26s %8s %-26s %6s %11s %9s %6s %9s\n" % ("case", "systems", "kernel", "calls", "avg us",
                                                              "GB/s", "frac", "HBM/alg"))
        for r in cases:
            f.write("%-26s %8d %-26s %6d %11.2f %9.0f %6.3f %9s\n" % (
                "n=%d m=%d N=%d%s" % (r["n"], r["m"], r["N"], " LTV" if r["ltv"] else ""), r["systems"],
                r.get("kernel", "?")[:26], r.get("calls", 0), r.get("average_ns", 0) / 1e3,
                r.get("achieved_GBps", 0), r.get("frac_of_8TBps", 0),
                "%.3f" % r["traffic_over_algorithmic"] if "traffic_over_algorithmic" in r else "-"))
    print(open(os.path.join(prof, tag + "_fill_profile.txt")).read())


if __name__ == "__main__":
    main()
