"""Per-launch durations out of a rocprofv3 --kernel-trace directory.

rocprofv3's --stats averages every launch of a run, the first ones included -- and the first
launches after idle run up to 15 % slower than the ones that follow (the clocks settle over 15-25
ms of load, tools/launch_series.py).  The profile scripts therefore let a run warm the device
first and read the steady state from the trace itself: the mean of the LAST `last` launches."""
import csv
import glob
import os


def durations(path, match):
    """Durations (ns) of the launches whose kernel name contains `match`, in start order."""
    rows = []
    for f in glob.glob(os.path.join(path, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return [d for _, d in sorted(rows)]


def steady_ns(path, match, last):
    d = durations(path, match)[-last:]
    return sum(d) / len(d) if d else None
