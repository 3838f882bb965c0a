#!/usr/bin/env python3
"""The finest pass at 32769 x 32769 on two physical placements of the same arrays in one process, for PMC runs.

State A: x (two arrays), b and the level-1 arrays cut out of ONE allocation, the process's first.  State B:
that allocation freed, `--spacer` GiB allocated in its place, and the same pool allocated again (same virtual
address, other physical memory).  Each state launches the pass `--launches` times; HIP-event times are printed,
and under `rocprofv3 --pmc ...` the counter rows of the mg_fused_kernel launches come in the same order
(first half state A, second half state B)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spacer", type=int, nargs="+", default=[40], help="GiB held while the pool is allocated again (one state per value)")
    ap.add_argument("--launches", type=int, default=4)
    ap.add_argument("--exp", type=int, default=14)
    args = ap.parse_args()
    import torch
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    n = args.exp
    N = 2 ** (n + 1) + 1
    plan = make_plan(N, N, n - 4, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    na, nc = L.ny * L.ld, Lc.ny * Lc.ld
    norm = ops.norm_slots()
    spacer = None
    for state, gib in enumerate([0] + args.spacer):
        if gib:
            spacer = torch.empty(gib << 27, dtype=torch.float64, device="cuda")
        pool = torch.zeros(3 * na + 2 * nc + 1024, dtype=torch.float64, device="cuda")
        off, arrs = 0, []
        for cnt, (ny, ld) in ((na, (L.ny, L.ld)),) * 3 + ((nc, (Lc.ny, Lc.ld)),) * 2:
            arrs.append(pool[off : off + cnt].view(ny, ld))
            off = (off + cnt + 15) // 16 * 16
        xa, xb, b, crs, cb = arrs
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        for k in range(args.launches):
            ops.smooth_fused(b, (xa, xb)[k & 1], (xb, xa)[k & 1], None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6,
                             L.h, norm, cb, Lc.nx, Lc.t0, Lc.t1)
            if k == 0:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        print(f"state {state}: {gib} GiB held, pool at GiB {pool.data_ptr() / 2 ** 30:.3f}: "
              f"{ev[0].elapsed_time(ev[1]) / (args.launches - 1):.3f} ms per pass", flush=True)
        del pool, arrs, xa, xb, b, crs, cb, spacer
        spacer = None
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
