#!/usr/bin/env python3
"""The finest pass and the 9-stage ascent pass at 32769 x 32769 for tile heights from 64 rows to a sixth of the
grid (one wave per strip and sixth: a single round of workgroups, no rows streamed twice)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from multigrid_amd._lib import check
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    n = 14
    N = 2 ** (n + 1) + 1
    plan = make_plan(N, N, n - 4, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    norm = ops.norm_slots()
    crs, cb = ops.zeros(Lc.ny, Lc.ld), ops.zeros(Lc.ny, Lc.ld)
    skip = [ops.zeros(L.ny, L.ld) for _ in range(int(os.environ.get("SKIP", "11")))]  # the first allocations are the slow kind
    b, xa, xb = (ops.zeros(L.ny, L.ld) for _ in range(3))
    for a in (b, xa, xb):
        a.normal_()

    def t(fn, reps=3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        fn()
        ev[0].record(ops.stream)
        for _ in range(reps):
            fn()
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    rows_list = [int(v) for v in sys.argv[1:]] or [0, 64, 128, 256, 512, 1024, 2048, 2731, 4096, 5462]
    for rows in rows_list:
        for lock in [int(v) for v in os.environ.get("LOCKSTEP", "0").split(",")]:
            check(ops.L.mg_set_tuning(b"fused_lockstep", lock))
            check(ops.L.mg_set_fused_tuning(1, rows))
            fin = t(lambda: ops.smooth_fused(b, xa, xb, None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6, L.h, norm,
                                             cb, Lc.nx, Lc.t0, Lc.t1))
            asc = t(lambda: ops.smooth_fused(b, None, xb, None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 9, L.h, None,
                                             prolong_after=3))
            des = t(lambda: ops.smooth_fused(b, None, None, None, None, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 3, L.h, None,
                                             cb, Lc.nx, Lc.t0, Lc.t1))
            print(f"rows per tile {rows:5d}, lockstep {lock}: finest pass {fin:.3f} ms, 9-stage ascent {asc:.3f} ms, "
                  f"descent without x {des:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
