#!/usr/bin/env python3
"""Kernels of one V-cycle of one rank from a rocprofv3 kernel trace of tools/rank_rehearsal.py (or of a rank of a
multi-GPU run): start relative to the end of the previous finest interior launch, duration, name, grid, queue.

    python tools/rank_cycle_timeline.py <kernel_trace.csv> [which cycle, counted from the end; default 2]
"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = max(int(r["Grid_Size_X"]) for r in rows if "mg_fused_kernel<6, true, false, true, true" in r["Kernel_Name"])
fin = [i for i, r in enumerate(rows) if "mg_fused_kernel<6, true, false, true, true" in r["Kernel_Name"] and int(r["Grid_Size_X"]) == big]
spans = [k for k in range(len(fin) - 1) if fin[k + 1] - fin[k] > 25]
k = spans[-(int(sys.argv[2]) if len(sys.argv) > 2 else 2)]
a, b = fin[k], fin[k + 1]
t0 = int(rows[a]["End_Timestamp"])
busy, idle = t0, 0
for r in rows[a + 1 : b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy:
        idle += s - busy
    busy = max(busy, e)
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:66]:66s} grid={r['Grid_Size_X']:>8s} q={r.get('Queue_Id')}")
print(f"cycle: {(busy - t0) / 1e3:.1f} us from the end of the previous finest interior launch, {idle / 1e3:.1f} us with no kernel running")
