#!/usr/bin/env python3
"""Time the finest-level streaming kernels under different tuning knobs (interleaved rounds in one
process, HIP events on the launch stream).  Usage: python tools/tune_stencil.py [--n 14]"""
import argparse
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=14)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--rows", type=int, nargs="*", default=[0, 32, 64, 256])
    ap.add_argument("--unroll", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--maps", type=int, nargs="*", default=[0, 1])
    args = ap.parse_args()
    import torch
    from multigrid_amd import _lib
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps

    N = 2 ** (args.n + 1) + 1
    plan = make_plan(N, N, 2, 1, 0, 1.0)
    L0, L1 = plan.levels[0], plan.levels[1]
    ops = HipOps()
    lib = _lib.lib()
    x = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    y = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    b = torch.rand((L0.ny, L0.ld), dtype=torch.float64, device="cuda")
    c = torch.rand((L1.ny, L1.ld), dtype=torch.float64, device="cuda")
    slots = ops.norm_slots()
    pts = (L0.t1 - L0.t0) * (L0.nx - 2)
    torch.cuda.synchronize()
    ops.begin()

    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record(ops.stream)
        for _ in range(args.iters):
            fn()
        e1.record(ops.stream)
        ops.stream.synchronize()
        return e0.elapsed_time(e1) / args.iters

    kernels = {
        "jacobi": (24, lambda: ops.jacobi(b, x, y, L0.nx, L0.t0, L0.t1, 0.5, True)),
        "resid+norm": (24, lambda: ops.residual(y, x, b, L0.nx, L0.t0, L0.t1, 0.5, slots, True)),
        "jac_zero": (16, lambda: ops.jacobi_from_zero(b, y, L0.nx, L0.t0, L0.t1, 0.5, True)),
    }
    configs = list(itertools.product([2], args.unroll, args.rows, args.maps))
    results = {}
    for rnd in range(args.rounds):
        for cfg in configs:
            v, u, rows, lm = cfg
            for k, val in (("vector", v), ("unroll", u), ("rows", rows), ("linear_map", lm)):
                _lib.check(lib.mg_set_tuning(k.encode(), val))
            for name, (bpp, fn) in kernels.items():
                results.setdefault((cfg, name), []).append(timed(fn))
        print(f"round {rnd} done", flush=True)
    print(f"grid {N}x{N}, {pts} points/launch; GB/s = algorithmic bytes / median time")
    print(f"{'vec':>3} {'unr':>3} {'rows':>4} {'lin':>3} | " + " | ".join(f"{k:>22}" for k in kernels))
    for cfg in configs:
        row = []
        for name, (bpp, _) in kernels.items():
            ts = sorted(results[(cfg, name)])
            med = ts[len(ts) // 2]
            row.append(f"{med:7.3f} ms {bpp * pts / med / 1e6:8.0f} GB/s")
        print(f"{cfg[0]:>3} {cfg[1]:>3} {cfg[2]:>4} {cfg[3]:>3} | " + " | ".join(row), flush=True)
    # transfer kernels (no knobs yet)
    t_r = timed(lambda: ops.restrict(x, L0.nx, c, L1.nx, L1.t0, L1.t1, 0, -1))
    t_p = timed(lambda: ops.prolong_add(c, L1.nx, x, L0.nx, L0.t0, L0.t1, 0, -1))
    print(f"restrict {t_r:7.3f} ms {10 * pts / t_r / 1e6:8.0f} GB/s | prolong {t_p:7.3f} ms {18 * pts / t_p / 1e6:8.0f} GB/s")


if __name__ == "__main__":
    main()
