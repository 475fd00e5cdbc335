"""Kernel-by-kernel timeline of the LAST BA iteration in a rocprofv3 --kernel-trace CSV: duration, gap to the previous
kernel, grid.   python3 tools/trace_iteration.py <dir with *_kernel_trace.csv> [anchor kernel substring]"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "ba_linearize"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
prev_end, tot, gaps = None, 0.0, 0.0
for r in rows[a:b]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev_end) if prev_end else 0
    name = r["Kernel_Name"].split("(")[0].replace("sfm::", "").replace("void ", "")[:34]
    print("%-36s dur %8.2f us  gap %6.2f  grid %s" % (name, (en - st) / 1e3, gap / 1e3, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "")))
    tot += (en - st) / 1e3; gaps += gap / 1e3
    prev_end = en
print("kernels %.1f us + gaps %.1f us = %.1f us" % (tot, gaps, tot + gaps))
