#!/usr/bin/env python3
"""Does the rate of the finest pass on a given trio of arrays change when OTHER allocations are released or made?
Allocates `--count` finest-size arrays, times the pass on the last three, releases the others, times again,
allocates them anew, times again; and the same for the first three."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=14)
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
    norm = ops.norm_slots()
    crs, cb = ops.zeros(Lc.ny, Lc.ld), ops.zeros(Lc.ny, Lc.ld)

    def t(b, xi, xo, reps=2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        for rep in range(reps + 1):
            ops.smooth_fused(b, xi, xo, None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6, L.h, norm, cb, Lc.nx,
                             Lc.t0, Lc.t1)
            if rep == 0:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def free_gib():
        return torch.cuda.mem_get_info()[0] / 2 ** 30

    A = [ops.zeros(L.ny, L.ld) for _ in range(args.count)]
    last, first = A[-3:], A[:3]
    print(f"{args.count} arrays allocated, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms, first three {t(*first):.3f} ms", flush=True)
    mid = A[3:-3]
    del A, mid
    torch.cuda.empty_cache()
    print(f"middle ones released, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms, first three {t(*first):.3f} ms", flush=True)
    mid = [ops.zeros(L.ny, L.ld) for _ in range(args.count - 6)]
    print(f"middle ones allocated again, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms, first three {t(*first):.3f} ms", flush=True)
    print("  one of the new ones as output:", f"{t(first[0], first[1], mid[0]):.3f} {t(last[0], last[1], mid[-1]):.3f} ms")
    del first
    torch.cuda.empty_cache()
    print(f"first three released, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms", flush=True)
    del mid
    torch.cuda.empty_cache()
    print(f"all but the last three released, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms", flush=True)
    small = [ops.zeros(Lc.ny, Lc.ld) for _ in range(12)]
    print(f"12 level-1-size arrays allocated, free {free_gib():.0f} GiB: last three {t(*last):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
