#!/usr/bin/env python3
"""The sweep of the reference's benchmark driver (test/test.py:33-94) on the GPU drop-in.

For n = 4..8 and n_para = n//3 .. n-1: the eight-vortex right-hand side split over four blocks,
ten warm-started solves with the driver's in-place mutation of the blocks before each, then one
table line per configuration: max relative error of the last solve, time per grid point, time
per solve, V-cycles of the last solve.  `--layout block` (default) runs the reference-exact
four-block `Multigrid`; `--layout global` runs `GlobalMultigrid` (the slab engine) on the stitched
right-hand side.  The last column checks the gathered solution: max|b - A x| over the whole grid,
relative to max|b|.

    python tools/harness.py [--layout block|global] [--nmin 4] [--nmax 8]
"""
import argparse
import os
import sys
from time import perf_counter

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

B_MAX = 50.0


def vortex_rhs(n):
    N = 2 ** (n + 1) + 1
    ax = np.linspace(-10, 10, N)
    X, Y = np.meshgrid(ax, ax)
    b0 = np.zeros_like(X)
    sign = 1.0
    for xv, yv in zip([4, -2, 6, 3, 3, 5, -5, -7], [-2, 6, 6, -5, 5, -5, -2, -2]):
        b0 += sign * B_MAX * np.exp(-((X - xv) ** 2 + (Y - yv) ** 2) * 7)
        sign = -sign
    return b0, ax[1] - ax[0]


def run_block(n, n_para, solves):
    from multigrid_amd.parallel_multigrid import BlockWorld, Multigrid, gather_blocks, residual, split
    b0, h = vortex_rhs(n)
    m = 2 ** n + 2
    world = BlockWorld()
    world.verbose = False
    b = [np.zeros((m, m)) for _ in range(4)]
    a = [np.zeros((m, m)) for _ in range(4)]
    R = [np.zeros((m, m)) for _ in range(4)]
    for rank in range(4):
        split(b0, b[rank], rank)
    mg = [Multigrid(b[k], a[k], R[k], h, B_MAX * 0.0005, n, n_para, rank=k, world=world) for k in range(4)]
    t = 0.0
    for _ in range(solves):
        for k in range(4):
            b[k][1:-1] += b[k][0:-2]
            b[k][:] /= 2
        t0 = perf_counter()
        err = max(mg[k].solve() for k in range(4))
        t += perf_counter() - t0
    a_full, b_full = np.zeros_like(b0), np.zeros_like(b0)
    gather_blocks(world, a, a_full)
    gather_blocks(world, b, b_full)
    r_full = np.zeros_like(b0)
    residual(r_full, a_full, b_full, h)
    return mg[0].it, err, t, np.abs(r_full[1:-1, 1:-1]).max() / np.abs(b_full).max()


def run_global(n, n_para, solves):
    from multigrid_amd.parallel_multigrid import GlobalMultigrid, residual
    b0, h = vortex_rhs(n)
    b = b0.copy()
    b[0] = b[-1] = 0
    b[:, 0] = b[:, -1] = 0
    a, R = np.zeros_like(b), np.zeros_like(b)
    mg = GlobalMultigrid(b, a, R, h, B_MAX * 0.0005, n, n_para)
    t = 0.0
    for _ in range(solves):
        b[1:-1] += b[0:-2]
        b[:] /= 2
        b[0] = b[-1] = 0
        t0 = perf_counter()
        err = mg.solve()
        t += perf_counter() - t0
    r_full = np.zeros_like(b)
    residual(r_full, a, b, h)
    return mg.it, err, t, np.abs(r_full[1:-1, 1:-1]).max() / np.abs(b).max()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layout", default="block", choices=["block", "global"])
    ap.add_argument("--nmin", type=int, default=4)
    ap.add_argument("--nmax", type=int, default=8)
    ap.add_argument("--solves", type=int, default=10)
    args = ap.parse_args()
    run = run_block if args.layout == "block" else run_global
    print(f"{'n':<2}{'n_para':>8}{'max_rel_er':>12}{'t_per_point':>14}{'t_solve':>11}{'v_cycles':>10}{'check |r|/|b|':>16}")
    for n in range(args.nmin, args.nmax + 1):
        for n_para in range(max(1, n // 3), n):
            it, err, t, chk = run(n, n_para, args.solves)
            pts = (2 ** (n + 1) + 1) ** 2
            print(f"{n:<2}{n_para:>8}{err / B_MAX:>12.1e}{t / (args.solves * pts):>14.2e}{t / args.solves:>11.2e}{it:>10}{chk:>16.1e}",
                  flush=True)


if __name__ == "__main__":
    main()
