#!/usr/bin/env python3
"""Turn the two rocprofv3 --pmc passes over tools/pmc_probe.py into HBM bytes per launch.

    python tools/pmc_summary.py fetch_counter_collection.csv write_counter_collection.csv [--n 14]
                                [--cycle-fetch csv --cycle-write csv] > profiles/traffic.json

The optional second pair are the same two passes over `tools/pmc_probe.py --cycle`: the counters of every kernel of
one whole V-cycle (the second of three: from after the first launch of the cycle's finest pass to the second) added up.

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts half of the bytes of
16-byte-per-lane streaming reads, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
Values are averaged over the launches of each kernel."""
import argparse
import csv
import json
import re
import sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(list)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            if name.startswith("void mg_") or name.startswith("mg_"):
                acc[re.sub(r"\(.*$", "", name).strip()].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


FINEST = "mg_fused_kernel<6, true, false, true, true"


def one_cycle(path, counter):
    """Sum of the counter over the dispatches of the second V-cycle, and their number."""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and "mg_" in row["Kernel_Name"]:
                rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if FINEST in r[1]]
    assert len(ends) >= 2, "the trace holds fewer than two cycles"
    part = rows[ends[0] + 1 : ends[1] + 1]
    return sum(r[2] for r in part), len(part)


# bytes per updated finest-level point: (per-operation compulsory bytes x the operations of the launch,
# what the launch itself must move)
BYTES = {
    "void mg_stencil5_kernel<0, 2, 1, true>": (24, 24),
    "void mg_stencil5_kernel<3, 2, 1, true>": (24, 24),
    "void mg_stencil5_kernel<1, 2, 1, true>": (16, 16),
    "void mg_restrict_kernel<true>": (10, 10),
    "void mg_prolong_kernel<true, true>": (18, 18),
    "void mg_fused_kernel<6, true, false, true, false, 1, 0, false, true>": (18 + 6 * 24 + 24, 8 + 2 + 8 + 8 + 8),
    "void mg_fused_kernel<3, false, true, true, false, 1, 0, false, true>": (16 + 2 * 24 + 24, 8 + 8 + 8),
    "void mg_fused_kernel<6, true, false, false, false, 1, 0, false, true>": (18 + 6 * 24, 8 + 2 + 8 + 8),
    "void mg_fused_kernel<6, true, false, true, true, 1, 0, false, true>": (18 + 6 * 24 + 24 + 10, 8 + 2 + 8 + 8 + 2),  # r not stored
    "void mg_fused_kernel<3, false, true, true, true, 1, 0, false, true>": (16 + 2 * 24 + 24 + 10, 8 + 2),  # x not stored
    "void mg_fused_kernel<9, true, true, false, false, 1, 3, false, true>": (16 + 2 * 24 + 18 + 6 * 24, 8 + 2 + 8),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--n", type=int, default=14)
    ap.add_argument("--cycle-fetch")
    ap.add_argument("--cycle-write")
    args = ap.parse_args()
    pts = (2 ** (args.n + 1) - 1) ** 2
    fetch = per_kernel(args.fetch_csv, "FETCH_SIZE")
    write = per_kernel(args.write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        e = {"fetch_bytes": 2 * 1024 * fetch[k], "write_bytes": 1024 * write[k]}
        e["hbm_bytes"] = e["fetch_bytes"] + e["write_bytes"]
        if k in BYTES:
            e["algorithmic_bytes"] = BYTES[k][0] * pts
            e["own_compulsory_bytes"] = BYTES[k][1] * pts
            e["hbm_over_own"] = e["hbm_bytes"] / e["own_compulsory_bytes"]
        kernels[k] = e

    def total(name):
        return kernels[name]["hbm_bytes"] if name in kernels else None

    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from multigrid_amd._lib import sources_digest
    out = {
        "csrc_sha256": sources_digest(),
        "jacobi_finest_bytes_per_launch": total("void mg_stencil5_kernel<0, 2, 1, true>"),
        "fused_finest_bytes_per_launch": total("void mg_fused_kernel<6, true, false, true, true, 1, 0, false, true>"),
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/pmc_probe.py, "
                f"{2 ** (args.n + 1) + 1} squared; FETCH_SIZE (KiB) doubled per the gfx950 correction for 16-B/lane "
                "streaming reads, WRITE_SIZE (KiB) as is; averages over the launches of each kernel",
        "kernels": kernels,
    }
    if args.cycle_fetch and args.cycle_write:
        f, nk = one_cycle(args.cycle_fetch, "FETCH_SIZE")
        w, nk2 = one_cycle(args.cycle_write, "WRITE_SIZE")
        assert nk == nk2, (nk, nk2)
        out["cycle"] = {"fetch_bytes": 2 * 1024 * f, "write_bytes": 1024 * w, "hbm_bytes": 2 * 1024 * f + 1024 * w, "kernels": nk,
                        "note": "every kernel of one V-cycle of bench.py's default workload (tools/pmc_probe.py --cycle, launched "
                                "kernel by kernel, no placement selection), same corrections; the small kernels' access widths are "
                                "not calibrated, they carry under 2 % of the bytes"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
