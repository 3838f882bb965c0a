#!/usr/bin/env python3
"""Prints one V-cycle of one rank of a multi-rank run from its rocprofv3 kernel trace (csv, see
tools/rank_timeline.sh): start relative to the interior launch of the finest pass, duration, name, grid.
A pass that was cut around its halo exchange shows as one large launch (interior rows, enqueued before the
exchange is set up) followed by two small ones (the rows at the slab ends, after the halo rows arrived).

    python tools/rank_timeline.py gpurun_out/prof_ranks/*/*_kernel_trace.csv
"""
import csv
import sys

FIN = "mg_fused_kernel<6, true, false, true, true"
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mg_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a cut finest pass: a large launch followed by two small ones of the same kernel
trip = [i for i, r in enumerate(rows) if FIN in r["Kernel_Name"] and i + 2 < len(rows) and FIN in rows[i + 1]["Kernel_Name"]
        and int(rows[i + 1]["Grid_Size_X"]) < int(r["Grid_Size_X"]) // 4]
if len(trip) < 3:
    sys.exit("no cut finest passes in this trace")
a, b = trip[-3], trip[-2]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a : b + 3]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:60]:60s} grid={r['Grid_Size_X']:>8s} queue={r['Queue_Id']}")
