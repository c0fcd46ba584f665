#!/usr/bin/env python3
"""profiles/<tag>_c4_profile.txt from tools/profile_c4.sh's rocprofv3 directories: per kernel of
the staged pipeline on C4 (B = 1024 per launch) the average duration, the share of the step, the
matrix-core busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, both summed over the chip by
rocprofv3), executed MFMA flops per launch and what that is against the measured and nominal
fp64 peaks, and the HBM bytes the counters saw."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kname(full):
    m = re.search(r"(compose_rowsets_kernel|hessian_gemm_kernel|hessian_kernel|gradient_kernel|"
                  r"constraints_kernel|resident_\w+|fused_assemble_kernel)", full)
    return m.group(1) if m else None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    stats = {}
    for f in glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = kname(r["Name"])
            if k:
                stats[k] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
    counters = {}
    for sub in ("mfma", "fetch", "write"):
        for f in glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = kname(r["Kernel_Name"])
                if k:
                    counters.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    med = lambda v: sorted(v)[len(v) // 2]
    lines = [open(os.path.join(src, "plain.txt")).read().strip().splitlines()[-1], ""]
    lines.append("%-26s %6s %9s %7s %13s %10s %8s %8s %9s %9s" % (
        "kernel (B=1024 / launch)", "calls", "avg ms", "% time", "MFMA insts", "TFLOP/s", "of 78.6", "of 47", "HBM MB", "HBM TB/s"))
    raw = []
    for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
        c = {n: med(v) for n, v in counters.get(k, {}).items()}
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / c["SQ_BUSY_CYCLES"] if c.get("SQ_BUSY_CYCLES") else 0
        # v_mfma_f64_16x16x4: 2 * 16 * 16 * 4 = 2048 flop per wave-instruction
        flops = c.get("SQ_INSTS_MFMA", 0) * 2048
        tf = flops / avg / 1e3 if avg else 0
        hbm = (c.get("WRITE_SIZE", 0) + 2 * c.get("FETCH_SIZE", 0)) * 1024 / 1e6
        lines.append("%-26s %6d %9.3f %7.1f %13.0f %10.1f %8.3f %8.3f %9.1f %9.2f" % (
            k, calls, avg / 1e6, pct, c.get("SQ_INSTS_MFMA", 0), tf, tf / 78.6, tf / 47.0, hbm,
            hbm / (avg / 1e6) / 1e3 if avg else 0))
        raw.append("%s: SQ_VALU_MFMA_BUSY_CYCLES %.0f  SQ_BUSY_CYCLES %.0f  SQ_WAVE_CYCLES %.0f  SQ_INSTS_VALU %.0f"
                   % (k, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0),
                      c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_INSTS_VALU", 0)))
    lines += ["", "TFLOP/s = SQ_INSTS_MFMA x 2048 flop (v_mfma_f64_16x16x4_f64, executed: only block pairs bi <= bj "
              "of a symmetric P) / average duration;", "78.6 = nominal fp64 matrix peak of gfx950, 47 = what "
              "tools/microbench/fp64_rate.hip sustains with this instruction.", "HBM MB = (WRITE_SIZE + 2 "
              "FETCH_SIZE) KiB, counter passes of their own (MI355X_MICROARCH.md, HBM).", "raw counters (median "
              "launch, summed over the chip by rocprofv3):"] + raw
    text = "\n".join(lines) + "\n"
    open(os.path.join(ROOT, "profiles", tag + "_c4_profile.txt"), "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
