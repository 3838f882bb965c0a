#!/usr/bin/env python3
"""Is the placement effect a property of single arrays or of their combination?  Allocates `--count` separate
32769 x 32784 float64 arrays, times a read-only stream (mg_maxabs), a write-only stream (mg_zero) and a
read+write pass (the first Jacobi sweep from zero: reads one array, writes another) on each, then times the
finest fused pass on trios picked by those single-array rates."""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=12)
    ap.add_argument("--exp", type=int, default=14)
    args = ap.parse_args()
    import torch
    from multigrid_amd._lib import check
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    n = args.exp
    N = 2 ** (n + 1) + 1
    plan = make_plan(N, N, n - 4, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    norm = ops.norm_slots()
    arrs = [ops.zeros(L.ny, L.ld) for _ in range(args.count)]
    crs, cb = ops.zeros(Lc.ny, Lc.ld), ops.zeros(Lc.ny, Lc.ld)

    def timed(fn, reps=5):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        fn()
        ev[0].record(ops.stream)
        for _ in range(reps):
            fn()
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    rd, wr = [], []
    for k, a in enumerate(arrs):
        t_r = timed(lambda: check(ops.L.mg_maxabs(a.data_ptr(), L.ny, L.nx, L.ld, norm.data_ptr(), ops.stream.cuda_stream)))
        t_w = timed(lambda: ops.zero_(a))
        rd.append(t_r), wr.append(t_w)
        print(f"array {k:2d} at GiB {a.data_ptr() / 2 ** 30:.3f}: read {t_r:.3f} ms ({L.ny * L.nx * 8 / t_r / 1e9:.2f} TB/s), "
              f"memset {t_w:.3f} ms", flush=True)

    def fused(ib, ia, io):
        return timed(lambda: ops.smooth_fused(arrs[ib], arrs[ia], arrs[io], None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi,
                                              L.oj, 6, L.h, norm, cb, Lc.nx, Lc.t0, Lc.t1), reps=3)

    order = sorted(range(args.count), key=lambda k: rd[k])
    print("by read time:", order)
    trios = [tuple(order[:3]), tuple(order[-3:]), (order[0], order[1], order[-1]), (order[-1], order[-2], order[0])]
    trios += [tuple(range(k, k + 3)) for k in range(0, args.count - 2, 3)]
    for t in trios:
        print(f"finest pass on arrays b={t[0]} x_in={t[1]} x_out={t[2]}: {fused(*t):.3f} ms", flush=True)
    # the single-operation kernels with an early and a late array as their output
    lo, hi = 2, args.count - 1
    for name, fn in (("jacobi sweep", lambda o: ops.jacobi(arrs[0], arrs[1], arrs[o], L.nx, L.t0, L.t1, L.h, True)),
                     ("sweep from zero", lambda o: ops.jacobi_from_zero(arrs[0], arrs[o], L.nx, L.t0, L.t1, L.h, True)),
                     ("residual", lambda o: ops.residual(arrs[o], arrs[1], arrs[0], L.nx, L.t0, L.t1, L.h, norm, True)),
                     ("9-stage ascent", lambda o: ops.smooth_fused(arrs[0], None, arrs[o], None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo,
                                                                   L.fix_hi, L.oj, 9, L.h, None, prolong_after=3)),
                     ("descent, x stored", lambda o: ops.smooth_fused(arrs[0], None, arrs[o], None, None, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo,
                                                                      L.fix_hi, L.oj, 3, L.h, None, cb, Lc.nx, Lc.t0, Lc.t1))):
        print(f"{name}: output array {lo}: {timed(lambda: fn(lo), 3):.3f} ms, output array {hi}: {timed(lambda: fn(hi), 3):.3f} ms", flush=True)
    # the same trio in every role assignment
    t = tuple(range(3))
    for perm in itertools.permutations(t):
        print(f"roles {perm}: {fused(*perm):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
