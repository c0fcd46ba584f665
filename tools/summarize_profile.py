#!/usr/bin/env python3
"""Condense a rocprofv3 run directory (as written on the GPU box under gpurun_out/)
into the small, tracked files of profiles/:

    python tools/summarize_profile.py gpurun_out/prof_r01 r01

expects   <dir>/stats/**/_kernel_stats.csv      (rocprofv3 --kernel-trace --stats)
          <dir>/fetch/**/_counter_collection.csv (rocprofv3 --pmc FETCH_SIZE, own pass)
          <dir>/write/**/_counter_collection.csv (rocprofv3 --pmc WRITE_SIZE, own pass)
          <dir>/bench_*.json                     (the bench lines of those runs)
writes    profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc.json and
          profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

HBM traffic follows MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request, so the read side is doubled.
"""
import csv
import glob
import json
import os
import sys

import proftrace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key in ("resident_spec_kernel", "resident_assemble_kernel", "fused_assemble_kernel",
                "toeplitz_scan_kernel", "ltv_sweep_kernel", "preview_blocked_kernel", "preview_staged_kernel",
                "lti_tables_small_kernel", "lti_tables_kernel", "compose_d_kernel",
                "fill_lti_quad_kernel", "fill_ltv_row_kernel", "fill_lti_tiny_kernel",
                "toeplitz_assemble_kernel", "tiled_assemble_kernel", "preview_direct_kernel", "goal_distance_kernel", "fill_lti_kernel", "fill_ltv_wave_kernel", "fill_ltv_kernel", "compose_rowsets_kernel", "hessian_kernel",
                "constraints_kernel", "compose_preview_kernel", "preview_kernel"):
        if key in name:
            return key
    return None


def counters(path, counter):
    out = {}
    for f in glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k and row["Counter_Name"] == counter:
                out.setdefault(k, []).append(float(row["Counter_Value"]))
    # the median ignores the first, cold launches
    return {k: sorted(v)[len(v) // 2] for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    src, tag = sys.argv[1], sys.argv[2]
    prof = os.path.join(ROOT, "profiles")
    os.makedirs(prof, exist_ok=True)
    rows = []
    for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Name"])
            rows.append([k or row["Name"][:60], row["Calls"], row["TotalDurationNs"],
                         row["AverageNs"], row["Percentage"], row["MinNs"], row["MaxNs"]])
    with open(os.path.join(prof, tag + "_kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "average_ns", "percent", "min_ns", "max_ns"])
        w.writerows(rows)
    fetch, nf = counters(os.path.join(src, "fetch"), "FETCH_SIZE")
    write, nw = counters(os.path.join(src, "write"), "WRITE_SIZE")
    bench = {}
    for name in ("stats", "fetch", "write"):
        p = os.path.join(src, "bench_%s.json" % name)
        if os.path.exists(p):
            bench[name] = json.load(open(p))
    batch = bench.get("stats", {}).get("config", {}).get("batch_per_gpu")
    summary = {"tag": tag, "command": "rocprofv3 ... -- python bench.py --no-cpu-baseline",
               "batch_per_gpu": batch, "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        fk, wk = fetch.get(k), write.get(k)
        summary["kernels"][k] = {
            "FETCH_SIZE_KiB_median": fk, "WRITE_SIZE_KiB_median": wk,
            "launches_sampled": [nf.get(k, 0), nw.get(k, 0)],
            "hbm_bytes_per_launch": (2 * fk if fk else 0) * 1024 + (wk or 0) * 1024,
            "note": "read side doubled (gfx950 FETCH_SIZE counts half of a streaming read)",
        }
    stats_avg = {r[0]: float(r[3]) for r in rows}
    summary["kernel_average_ns"] = {k: v for k, v in stats_avg.items() if short(k)}
    # (the average above covers every launch of the run, the first, slower ones included; the
    # timed region of bench.py is its last `steps` launches)
    for k in ("resident_spec_kernel", "resident_assemble_kernel"):
        steady = proftrace.steady_ns(os.path.join(src, "stats"), k, 2000)
        if steady:
            summary.setdefault("kernel_average_last_2000_launches_ns", {})[k] = steady
    if "stats" in bench:
        summary["bench_in_profiled_run"] = {
            "value": bench["stats"]["value"],
            "assemble_avg_launch_ms_hipEvent": bench["stats"]["roofline"]["avg_launch_ms"],
            "step": bench["stats"]["config"].get("step"),
        }
        if "fill_in_step" in bench["stats"]:
            summary["bench_in_profiled_run"]["fill_avg_launch_ms_hipEvent"] = \
                bench["stats"]["fill_in_step"]["avg_launch_ms"]
    json.dump(summary, open(os.path.join(prof, tag + "_pmc.json"), "w"), indent=1)
    dominant = "resident_spec_kernel" if "resident_spec_kernel" in summary["kernels"] \
        else "resident_assemble_kernel"
    if dominant in summary["kernels"]:
        fused = "fill_in_step" not in bench.get("stats", bench.get("fetch", {}))
        json.dump({"kernel": dominant, "batch_per_gpu": batch, "source": tag + "_pmc.json",
                   "k1_fused": fused,
                   "traffic_bytes_per_launch": summary["kernels"][dominant]["hbm_bytes_per_launch"]},
                  open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
