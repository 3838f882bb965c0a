#!/usr/bin/env python3
"""Time the fused finest-level passes under their tuning knobs (interleaved rounds, HIP events).
Usage: python tools/tune_fused.py [--n 14]"""
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
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--rows", type=int, nargs="*", default=[128, 256, 512, 1024])
    ap.add_argument("--unroll", type=int, nargs="*", default=[1, 2])
    ap.add_argument("--waves", type=int, nargs="*", default=[4], help="column strips (waves) per workgroup")
    ap.add_argument("--only", nargs="*", default=None, help="kernel names to time (default: all)")
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
    mk = lambda L: torch.rand((L.ny, L.ld), dtype=torch.float64, device="cuda")  # noqa: E731
    x, y, b, r, c, cb = mk(L0), mk(L0), mk(L0), mk(L0), mk(L1), mk(L1)
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

    def f(k, pro, zero, res, rs=False, xout=True, after=0):
        return lambda: ops.smooth_fused(b, None if zero else x, y if xout else None, r if res else None, c if pro else None,
                                        L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, k, 0.5,
                                        slots if (res or rs) else None, cb if rs else None, L1.nx, L1.t0, L1.t1,
                                        prolong_after=after)

    # name: (compulsory bytes/pt of the pass, algorithmic bytes/pt of the separate kernels, fn)
    kernels = {
        "P+3J": (26, 18 + 72, f(3, True, False, False)),
        "3J+R": (32, 72 + 24, f(3, False, False, True)),
        "3J": (24, 72, f(3, False, False, False)),
        "0+3J+R": (24, 16 + 48 + 24, f(3, False, True, True)),
        "P+6J+R": (34, 18 + 144 + 24, f(6, True, False, True)),
        "P+6J": (26, 18 + 144, f(6, True, False, False)),
        "P+4J": (26, 18 + 96, f(4, True, False, False)),
        "P+6J+RS": (28, 18 + 144 + 24 + 10, f(6, True, False, False, True)),
        "P+6J+R+RS": (36, 18 + 144 + 24 + 10, f(6, True, False, True, True)),
        "0+3J+RS": (18, 16 + 48 + 24 + 10, f(3, False, True, False, True)),
        "1J+RS": (26, 24 + 24 + 10, f(1, False, False, False, True)),
        "0+3J+RS-x": (10, 16 + 48 + 24 + 10, f(3, False, True, False, True, xout=False)),
        "0+3J+P+6J": (18, 16 + 48 + 18 + 144, f(9, True, True, False, after=3)),
    }
    if args.only:
        kernels = {k: v for k, v in kernels.items() if k in args.only}
    configs = list(itertools.product(args.unroll, args.rows, args.waves))
    results = {}
    for rnd in range(args.rounds):
        for cfg in configs:
            _lib.check(lib.mg_set_fused_tuning(cfg[0], cfg[1]))
            _lib.check(lib.mg_set_tuning(b"fused_waves", cfg[2]))
            for name, (_, _, fn) in kernels.items():
                results.setdefault((cfg, name), []).append(timed(fn))
        print(f"round {rnd} done", flush=True)
    print(f"grid {N}x{N}; per kernel: median ms, TB/s of the pass's own compulsory bytes")
    print(f"{'unr':>3} {'rows':>5} {'wv':>2} | " + " | ".join(f"{k:>14}" for k in kernels))  # noqa
    for cfg in configs:
        row = []
        for name, (bpp, _, _) in kernels.items():
            ts = sorted(results[(cfg, name)])
            med = ts[len(ts) // 2]
            row.append(f"{med:6.2f} {bpp * pts / med / 1e9:6.2f}")
        print(f"{cfg[0]:>3} {cfg[1]:>5} {cfg[2]:>2} | " + " | ".join(f"{v:>14}" for v in row), flush=True)


if __name__ == "__main__":
    main()
