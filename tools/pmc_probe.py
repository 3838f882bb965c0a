#!/usr/bin/env python3
"""A handful of full-size launches of the streaming kernels, for rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/pmc_probe.py [--n 14]

Counter collection serialises dispatches and is slow, so this probe launches each kernel of the
V-cycle's finest level a few times on a 32769 x 32769 grid and nothing else (no graph, no
right-hand-side generation: the arrays are filled by one torch call each).

    ... -- python3 tools/pmc_probe.py --cycle

launches whole V-cycles instead: the engine of bench.py's default workload (n_para = n - 4, dead whole-domain levels
computed every cycle) without graph and without placement selection, the opening pass and three cycles, kernel by
kernel; tools/pmc_summary.py adds up the middle cycle (the last one also stores the residual array)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=14)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--fused-rows", type=int, default=0, help="tile height of the fused passes (0 = automatic)")
    ap.add_argument("--cycle", action="store_true", help="whole V-cycles of the engine instead of single kernels")
    args = ap.parse_args()
    import torch
    if args.cycle:
        from multigrid_amd.engine import SlabMultigrid
        N = 2 ** (args.n + 1) + 1
        eng = SlabMultigrid(N, N, max(1, args.n - 4), 20.0 / 2 ** (args.n + 1), 0.0, world=1, rank=0, eager_dead_levels=True,
                            use_graph=False, tune_placement=False)
        b = torch.rand((N, N), dtype=torch.float64, device="cuda")
        b[0] = b[-1] = 0
        b[:, 0] = b[:, -1] = 0
        eng.set_b_rows(b)
        del b
        eng.presmooth()
        errs = eng.cycles(3)
        torch.cuda.synchronize()
        print(f"grid {N}x{N}, n_para {eng.n_para}: 3 cycles, max|r| {errs}", flush=True)
        return
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps

    N = 2 ** (args.n + 1) + 1
    plan = make_plan(N, N, 2, 1, 0, 1.0)
    L0, L1 = plan.levels[0], plan.levels[1]
    ops = HipOps()
    if args.fused_rows:
        from multigrid_amd import _lib
        _lib.check(_lib.lib().mg_set_fused_tuning(2, args.fused_rows))
    x = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    y = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    b = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    c = torch.rand((L1.ny, L1.ld), dtype=torch.float64, device="cuda")
    r = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    cb = torch.rand((L1.ny, L1.ld), dtype=torch.float64, device="cuda")
    slots = ops.norm_slots()
    torch.cuda.synchronize()
    ops.begin()
    for k in range(args.reps):
        ops.jacobi(b, x, y, L0.nx, L0.t0, L0.t1, 0.5, True)
        ops.residual(y, x, b, L0.nx, L0.t0, L0.t1, 0.5, slots, True)
        ops.restrict(x, L0.nx, c, L1.nx, L1.t0, L1.t1, 0, -1)
        ops.prolong_add(c, L1.nx, x, L0.nx, L0.t0, L0.t1, 0, -1)
        ops.jacobi_from_zero(b, y, L0.nx, L0.t0, L0.t1, 0.5, True)
        ops.smooth_fused(b, x, y, r, c, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6, 0.5, slots)
        ops.smooth_fused(b, None, y, r, None, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 3, 0.5, None)
        ops.smooth_fused(b, x, y, None, c, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6, 0.5, None)
        # the finest-level pass of the cycle, with the restriction fused in
        ops.smooth_fused(b, x, y, None, c, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6, 0.5, slots,
                         cb, L1.nx, L1.t0, L1.t1)
        # the recompute scheme of the large coarse levels: descent without x output, 9-sweep ascent
        ops.smooth_fused(b, None, None, None, None, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 3, 0.5, None,
                         cb, L1.nx, L1.t0, L1.t1)
        ops.smooth_fused(b, None, y, None, c, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 9, 0.5, None,
                         prolong_after=3)
        ops.synchronize()
        print(f"rep {k} done", flush=True)
    pts = (L0.t1 - L0.t0) * (L0.nx - 2)
    print(f"grid {N}x{N}: {pts} points per launch; algorithmic bytes: jacobi/residual {24 * pts}, "
          f"restrict {10 * pts}, prolong {18 * pts}, jacobi_from_zero {16 * pts}", flush=True)


if __name__ == "__main__":
    main()
