#!/usr/bin/env python3
"""Can the finest level's arrays be placed well WITHOUT timing anything?  In a fresh process: the coarse right-hand side
first, then b and r, then ballast up to about 96 GiB handed out, then one x array, ballast again, the other x array;
the ballast is freed.  Then the cycle's finest pass (32769 x 32769) is timed in both directions of x, as
SlabMultigrid._place_finest times its candidates.  Compare with --plain (the four arrays as the first allocations).

    python tools/placement_recipe.py [--ballast-a 10] [--ballast-b 11] [--plain]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ballast-a", type=int, default=10)
    ap.add_argument("--ballast-b", type=int, default=11)
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--exp", dest="n", type=int, default=14)
    args = ap.parse_args()
    import torch
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    N = 2 ** (args.n + 1) + 1
    plan = make_plan(N, N, 2, 1, 0, 20.0 / 2 ** (args.n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    ops = HipOps()
    cb = ops.zeros(Lc.ny, Lc.ld)
    if args.plain:
        xa, xb, b, r = (ops.zeros(L.ny, L.ld) for _ in range(4))
    else:
        b, r = ops.zeros(L.ny, L.ld), ops.zeros(L.ny, L.ld)
        ballast = [ops.zeros(L.ny, L.ld) for _ in range(args.ballast_a)]
        xa = ops.zeros(L.ny, L.ld)
        ballast += [ops.zeros(L.ny, L.ld) for _ in range(args.ballast_b)]
        xb = ops.zeros(L.ny, L.ld)
        del ballast
        torch.cuda.empty_cache()
    crs, norm = ops.zeros(Lc.ny, Lc.ld), ops.norm_slots()
    print("GiB addresses: cb %.1f b %.1f r %.1f xa %.1f xb %.1f crs %.1f" % tuple(t.data_ptr() / 2 ** 30 for t in (cb, b, r, xa, xb, crs)), flush=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def time_pass(x_in, x_out, reps=6):
        ops.begin()
        for rep in range(reps + 1):
            ops.smooth_fused(b, x_in, x_out, None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6, L.h, norm, cb, Lc.nx, Lc.t0,
                             Lc.t1)
            if rep == 0:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    t1, t2 = time_pass(xa, xb), time_pass(xb, xa)
    print(f"{'plain' if args.plain else f'recipe (ballast {args.ballast_a} + {args.ballast_b})'}: finest pass {t1:.3f} ms xa->xb, {t2:.3f} ms xb->xa, "
          f"mean {(t1 + t2) / 2:.3f} ms (zeroed arrays)", flush=True)


if __name__ == "__main__":
    main()
