#!/usr/bin/env python3
"""The steps of the engine's placement selection one by one, with the finest pass timed on the chosen arrays after
each: 13 candidates allocated; the others released; the remaining level arrays allocated."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    n = 14
    N = 2 ** (n + 1) + 1
    plan = make_plan(N, N, n - 4, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    norm = ops.norm_slots()
    crs0, cb0 = ops.zeros(Lc.ny, Lc.ld), ops.zeros(Lc.ny, Lc.ld)
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 13

    def t(b, xi, xo, reps=3, crs=None, cb=None):
        crs = crs0 if crs is None else crs
        cb = cb0 if cb is None else cb
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

    A = [ops.zeros(L.ny, L.ld) for _ in range(count)]
    xa, xb, b9, a0, a1 = A[-1], A[-2], A[-4], A[0], A[1]

    def report(label):
        print(f"{label}: free {torch.cuda.mem_get_info()[0] / 2 ** 30:.0f} GiB | inputs 0,1 -> {count - 2}: {t(a0, a1, xb):.3f} | "
              f"b={count - 4}, {count - 1} -> {count - 2}: {t(b9, xa, xb):.3f} | b={count - 4}, {count - 2} -> {count - 1}: {t(b9, xb, xa):.3f} | "
              f"b=0, {count - 1} -> {count - 2}: {t(a0, xa, xb):.3f} ms", flush=True)

    report(f"{count} candidates allocated")
    keep = {0, 1, count - 4, count - 2, count - 1}
    for k in range(count):
        if k not in keep:
            A[k] = None
    torch.cuda.empty_cache()
    report("the others released")
    small = [ops.zeros(Lc.ny, Lc.ld) for _ in range(6)]
    report("6 level-1-size arrays allocated")
    print("  the new small arrays as the correction (read) and the coarse right-hand side (written), b=9, 12 -> 11:",
          " ".join(f"{t(b9, xa, xb, crs=small[k], cb=small[k + 1]):.3f}" for k in range(5)), flush=True)
    print("  only the coarse right-hand side new:", " ".join(f"{t(b9, xa, xb, cb=small[k]):.3f}" for k in range(6)), flush=True)
    print("  only the correction new:", " ".join(f"{t(b9, xa, xb, crs=small[k]):.3f}" for k in range(6)), flush=True)
    big = [ops.zeros(L.ny, L.ld) for _ in range(4)]
    report("4 more finest-size arrays allocated")
    print("  one of those as the output:", f"{t(a0, a1, big[0]):.3f} {t(a0, a1, big[-1]):.3f} ms")


if __name__ == "__main__":
    main()
