#!/usr/bin/env python3
"""Prints the kernels of one V-cycle from a rocprofv3 kernel trace (csv): start relative to the end of
the previous finest pass, duration, name, grid.  Shows what runs beside what and where the stream idles.

    rocprofv3 --kernel-trace --output-format csv -d out -- python3 bench.py --no-cpu-baseline
    python tools/cycle_timeline.py out/*/*_kernel_trace.csv [index of the finest pass to start after]

Without an index it takes the third last span between two finest passes that holds a whole cycle (more than
20 kernels): the placement sampling and the untuned leg launch finest passes back to back, which are no cycles.
"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mg_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fin = [i for i, r in enumerate(rows) if "mg_fused_kernel<6, true, false, true, true" in r["Kernel_Name"]]
whole = [k for k in range(len(fin) - 1) if fin[k + 1] - fin[k] > 20]
k = int(sys.argv[2]) if len(sys.argv) > 2 else whole[-3]
a, b = fin[k], fin[k + 1]
t0 = int(rows[a]["End_Timestamp"])
busy_end = t0
idle = 0.0
for r in rows[a + 1 : b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        idle += s - busy_end
    busy_end = max(busy_end, e)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:64]:64s} grid={r['Grid_Size_X']}")
print(f"cycle: {(busy_end - t0) / 1e3:.1f} us from the end of the previous finest pass, {idle / 1e3:.1f} us with no kernel running")
