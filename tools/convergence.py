#!/usr/bin/env python3
"""Time and traffic to solution of the slab engine under the schedule options, one GPU.

For each grid size 2^(n+1)+1 (the eight-vortex right-hand side of the reference's driver,
test/test.py:40-58) and each tolerance, solve() is run from x = 0 with
  * the reference's schedule (whole-domain levels below the first never feed back,
    core/parallel_multigrid.py:302),
  * fix_coarse_ascent=True (the textbook ascent through every whole-domain level),
  * the same with damped Jacobi sweeps (jacobi_omega=0.8; the reference's undamped sweep leaves the
    checkerboard mode of the error untouched, which is the floor max|r| stalls at),
  * the same with red-black Gauss-Seidel sweeps on the fine levels (fine_smoother="redblack": one kernel per
    colour and operation, no fused passes, so fewer cycles but several times the traffic per cycle),
for several numbers of Jacobi levels, and one line is printed per run: V-cycles, whether the
stop rule ended on the tolerance, final max|r|, wall time of the solve, algorithmic bytes moved
(cycles x levels.algorithmic_bytes_per_cycle) and max|r| reduction per cycle (geometric mean).

    python tools/convergence.py [--exps 10 12 14] [--eps 0.025 2.5e-6]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exps", type=int, nargs="+", default=[10, 12, 14])
    ap.add_argument("--eps", type=float, nargs="+", default=[0.025, 2.5e-6])
    args = ap.parse_args()
    import torch

    from bench import fill_rhs
    from multigrid_amd.engine import SlabMultigrid
    from multigrid_amd.levels import algorithmic_bytes_per_cycle

    print(f"{'n':>3}{'n_para':>7}{'schedule':>12}{'eps':>9}{'cycles':>7}{'stop':>10}{'max|r|':>10}{'ms':>9}"
          f"{'GB (algorithmic)':>18}{'reduction/cycle':>17}", flush=True)
    for n in args.exps:
        for n_para in sorted({max(2, n - 8), n - 6, n - 4}):
            if n_para < 1 or n_para >= n:
                continue
            for name, fix, om, sm in (("reference", False, 1.0, "jacobi"), ("textbook", True, 1.0, "jacobi"),
                                      ("tb+damped", True, 0.8, "jacobi"), ("tb+redblack", True, 1.0, "redblack")):
                eng = SlabMultigrid(2 ** (n + 1) + 1, 2 ** (n + 1) + 1, n_para, 20.0 / 2 ** (n + 1), 0.0,
                                    fix_coarse_ascent=fix, eager_dead_levels=True, jacobi_omega=om, fine_smoother=sm)
                fill_rhs(eng, torch, n, 0)
                gb = algorithmic_bytes_per_cycle(eng.plan) / 1e9
                for eps in args.eps:
                    eng.epsilon = eps
                    eng.set_x_rows(0.0)
                    eng.solve()  # graphs captured, kernels loaded
                    eng.set_x_rows(0.0)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    err = eng.solve()
                    torch.cuda.synchronize()
                    ms = 1e3 * (time.perf_counter() - t0)
                    hist = eng.err_history
                    red = (hist[-1] / hist[0]) ** (1.0 / max(1, len(hist) - 1)) if len(hist) > 1 and hist[0] > 0 else float("nan")
                    stop = "eps" if err <= eps else ("diverged" if eng.diverged else "cap")
                    print(f"{n:>3}{n_para:>7}{name:>12}{eps:>9.1e}{eng.it:>7}{stop:>10}{err:>10.2e}"
                          f"{ms:>9.2f}{gb * eng.it:>18.1f}{red:>17.3f}", flush=True)
                eng.close()
                del eng
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
