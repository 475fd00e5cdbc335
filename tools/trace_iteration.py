"""Kernel-by-kernel timeline of the LAST BA iteration in a rocprofv3 --kernel-trace CSV: duration and grid of every
kernel, the iteration's period (start of its first kernel to the start of the next iteration's first kernel) and the
part of that period in which no kernel ran.  (rocprofv3 stamps a queued kernel's start at the end of its predecessor,
so per-kernel "gaps" read 0; only the period minus the kernels' sum sees the time between kernels.)
    python3 tools/trace_iteration.py <dir with *_kernel_trace.csv> [anchor kernel substring]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "ba_linearize"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
tot = 0.0
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("sfm::", "").replace("void ", "")[:34]
    print("%-36s dur %8.2f us  grid %s" % (name, (en - st) / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
    tot += (en - st) / 1e3
period = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
# periods of all complete iterations of the trace (median): one iteration can be stretched by the profiler itself
per = sorted((int(rows[j]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"])) / 1e3 for i, j in zip(idx[:-1], idx[1:]))
print("kernels %.1f us; iteration period %.1f us (median of %d: %.1f us); outside kernels %.1f us" % (tot, period, len(per), per[len(per) // 2], period - tot))
