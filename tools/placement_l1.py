#!/usr/bin/env python3
"""Level 1 of a 32769 x 32769 engine: its two passes (descent without x into b[2]; 9-stage ascent into xa[1]) timed
on the engine's own arrays and with freshly allocated arrays as their outputs: is there a slow and a fast kind of
output allocation for these passes too, and which one did set-up get?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from multigrid_amd.engine import SlabMultigrid
    n, n_para = 14, 10
    N = 2 ** (n + 1) + 1
    eng = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0, eager_dead_levels=True, tune_placement=True)
    bench.fill_rhs(eng, torch, n, 0)
    eng.presmooth()
    eng.cycles(2)
    ops, L1, L2 = eng.ops, eng.lv[1], eng.lv[2]

    def t(fn, reps=4):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        fn()
        ev[0].record(ops.stream)
        for _ in range(reps):
            fn()
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def ascent(out):
        return t(lambda: ops.smooth_fused(eng.b[1], None, out, None, eng._xa[2], L1.nx, L2.nx, L1.t0, L1.t1, L1.fix_lo, L1.fix_hi,
                                          L1.oj, 9, L1.h, None, prolong_after=3))

    def descent(cb):
        return t(lambda: ops.smooth_fused(eng.b[1], None, None, None, None, L1.nx, L2.nx, L1.t0, L1.t1, L1.fix_lo, L1.fix_hi, L1.oj,
                                          3, L1.h, None, cb, L2.nx, L2.t0, L2.t1))

    print(f"engine's own arrays: ascent into xa[1] {ascent(eng._xa[1]):.4f} ms, into xb[1] {ascent(eng._xb[1]):.4f} ms; "
          f"descent into b[2] {descent(eng.b[2]):.4f} ms")
    big = [ops.zeros(L1.ny, L1.ld) for _ in range(10)]
    print("ascent into fresh arrays:", " ".join(f"{ascent(a):.4f}" for a in big), flush=True)
    small = [ops.zeros(L2.ny, L2.ld) for _ in range(10)]
    print("descent into fresh arrays:", " ".join(f"{descent(a):.4f}" for a in small), flush=True)
    print("ascent into the finest arrays' memory (a view of r[0]):", f"{ascent(eng.r[0].view(-1)[: L1.ny * L1.ld].view(L1.ny, L1.ld)):.4f}")


if __name__ == "__main__":
    main()
