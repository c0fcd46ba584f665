#!/usr/bin/env python3
"""Condense gpurun_out/c4_<tag> (tools/profile_tiled.sh) into profiles/<tag>_c4_profile.txt:
per kernel of the tiled path -- calls, average duration, executed matrix-core flops, HBM traffic
(WRITE_SIZE + 2 FETCH_SIZE, MI355X_MICROARCH.md section HBM) against the algorithmic bytes.
    python tools/summarize_tiled_profile.py gpurun_out/c4_r03 r03 8192"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("toeplitz_assemble_kernel", "tiled_assemble_kernel", "lti_tables_kernel", "compose_d_kernel")


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


def trace(path):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
    return acc


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sorted(v)[len(v) // 2] for c, v in cs.items()} for k, cs in acc.items()}


def main():
    src, tag, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
    t = trace(os.path.join(src, "stats"))
    m = counters(os.path.join(src, "mfma"))
    fe = counters(os.path.join(src, "fetch"))
    wr = counters(os.path.join(src, "write"))
    no, nc, ng, npar = 384, 1536, 12, 108
    algo = 8 * (no * no + no + nc * no + nc) + 8 * (ng + npar + 144 + 72)
    lines = [open(os.path.join(src, "plain.txt")).read().strip(), ""]
    lines.append("kernel (B=%d in one call)        calls    avg ms   MFMA insts    TFLOP/s  of 78.6   HBM MB  HBM TB/s  x algorithmic" % B)
    total_ms = 0.0
    for k in KEYS:
        if k not in t:
            continue
        d = sorted(t[k])
        avg = sum(d[1:]) / max(len(d) - 1, 1) if len(d) > 1 else d[0]     # (the first call warms up)
        total_ms += avg
        mf = m.get(k, {}).get("SQ_INSTS_MFMA", 0.0)
        tf = mf * 2048 / (avg * 1e-3) / 1e12
        mb = (wr.get(k, {}).get("WRITE_SIZE", 0.0) + 2 * fe.get(k, {}).get("FETCH_SIZE", 0.0)) * 1024 / 1e6
        lines.append("%-32s %6d %9.3f %12.0f %10.1f %8.3f %8.1f %9.2f %10.2f" % (
            k, len(d), avg, mf, tf, tf / 78.6, mb, mb / 1e6 / (avg * 1e-3) if avg else 0.0,
            mb * 1e6 / (algo * B) if k.endswith("assemble_kernel") else 0.0))
    lines.append("")
    lines.append("all kernels of one call: %.3f ms -> %.3e assemblies/s; algorithmic bytes %d per assembly -> %.2f TB/s "
                 "(%.3f of 8 TB/s)" % (total_ms, B / total_ms * 1e3, algo, algo * B / total_ms / 1e9, algo * B / total_ms / 1e9 / 8))
    lines.append("TFLOP/s = SQ_INSTS_MFMA x 2048 flop (v_mfma_f64_16x16x4_f64, executed: structurally zero tiles and the "
                 "lower block pairs of the symmetric P are not multiplied) / average duration; 78.6 = nominal fp64 matrix peak.")
    lines.append("HBM MB = (WRITE_SIZE + 2 FETCH_SIZE) KiB, counter passes of their own (MI355X_MICROARCH.md, HBM).")
    lines.append("raw counters (median launch, summed over the chip by rocprofv3):")
    for k in KEYS:
        if k in m:
            lines.append("%s: %s" % (k, "  ".join("%s %.0f" % kv for kv in sorted(m[k].items()))))
    text = "\n".join(lines) + "\n"
    with open(os.path.join(ROOT, "profiles", "%s_c4_profile.txt" % tag), "w") as f:
        f.write(text)
    print(text)


if __name__ == "__main__":
    main()
