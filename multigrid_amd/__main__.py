"""`python -m multigrid_amd`: the script part of the reference's solver module
(core/parallel_multigrid.py:351-437) on the GPU drop-in.

A single Gaussian source on the (2^(n+1)+1)^2 grid over [-10, 10]^2 is cut into the four blocks,
solved twice (the second solve starts from the first one's solution), the reference's six lines are
printed per solve, the blocks are gathered and the residual of the gathered solution is evaluated
on the whole grid.  `--png FILE` writes the reference's two-panel figure (solution, residual / max b)
instead of opening a window; without it only the numbers are printed.

    python -m multigrid_amd [--n 8] [--n-para 4] [--png out.png]
"""
import argparse
from time import perf_counter

import numpy as np

from .parallel_multigrid import BlockWorld, Multigrid, gather_blocks, residual, split


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m multigrid_amd")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--n-para", type=int, default=4)
    ap.add_argument("--png", default=None, help="write the solution / residual figure to this file")
    args = ap.parse_args(argv)
    n, n_para = args.n, args.n_para
    b_max = 50
    epsilon = b_max * 0.0005

    nx1 = 2 ** n + 1
    nx0 = 2 ** (n + 1) + 1
    x = np.linspace(-10, 10, nx0)
    y = np.linspace(-10, 10, nx0)
    h = y[1] - y[0]
    X, Y = np.meshgrid(x, y)
    b0 = b_max * np.exp(-(X ** 2 + Y ** 2) * 4)

    world = BlockWorld()
    world.verbose = False
    b = [np.zeros((nx1 + 1, nx1 + 1)) for _ in range(4)]
    for rank in range(4):
        split(b0, b[rank], rank)
    a = [np.zeros_like(b[0]) for _ in range(4)]
    R = [np.zeros_like(b[0]) for _ in range(4)]
    poisson = [Multigrid(b[k], a[k], R[k], h, epsilon, n, n_para, rank=k, world=world) for k in range(4)]

    results = []
    for _ in range(2):
        t0 = perf_counter()
        m_err = max(p.solve() for p in poisson)
        t = perf_counter() - t0
        print(f"nx = {nx1}")
        print(f"n_para = {n_para}")
        print(f"m_err/b_max={m_err / b_max:.1e}")
        print(f"time {t / (2 ** (n + 1) - 1):.3e} s/point")
        print(f"time {t:.3e} s")
        print(f"v cycles {poisson[0].it}")
        results.append((poisson[0].it, m_err))

    a_full = np.zeros_like(b0)
    R_full = np.zeros_like(b0)
    gather_blocks(world, a, a_full)
    gather_blocks(world, R, R_full)
    residual(R_full, a_full, b0, h)
    r_max = np.amax(np.abs(R_full / b_max))
    print(f"residual of the gathered solution / max(B): {r_max:.1e}")

    if args.png:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, (ax0, ax1) = plt.subplots(1, 2)
        ax0.pcolormesh(a_full)
        ax1.pcolormesh(R_full / b_max, cmap="bwr", vmin=-r_max, vmax=r_max)
        fig.suptitle(f"{nx0}x{nx0} grid points")
        ax0.set_title("phi")
        ax1.set_title(f"Residual / max(B) max={r_max:.1e}")
        for ax in (ax0, ax1):
            ax.set_aspect("equal")
            ax.axis("off")
        fig.savefig(args.png, dpi=150)
        print(f"wrote {args.png}")
    return results, a_full, R_full


if __name__ == "__main__":
    main()
