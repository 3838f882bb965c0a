#!/usr/bin/env python3
"""Times the finest pass on an engine's own arrays right after construction (placement selected), after the
right-hand side was filled in, and after a few cycles: does what the selection measured still hold?"""
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
    eng = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0, eager_dead_levels=True,
                        tune_placement="verbose" if "--no-tune" not in sys.argv else False)
    print("placement:", eng.placement, f"{eng.placement_seconds:.2f} s")
    ops, L0, L1 = eng.ops, eng.plan.levels[0], eng.plan.levels[1]
    xs = [eng._xa[0], eng._xb[0]]

    def t(label, b=None, crs=None, cb=None, iters=6):
        b = eng.b[0] if b is None else b
        crs = eng._xa[1] if crs is None else crs
        cb = eng.b[1] if cb is None else cb
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        for k in range(iters + 2):
            ops.smooth_fused(b, xs[k & 1], xs[1 - (k & 1)], None, crs, L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6,
                             L0.h, eng.norm, cb, L1.nx, L1.t0, L1.t1)
            if k == 1:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        print(f"{label}: {ev[0].elapsed_time(ev[1]) / iters:.3f} ms", flush=True)

    print("addresses (GiB): xa %.2f xb %.2f b %.2f r %.2f | xa1 %.2f b1 %.2f" % tuple(
        a.data_ptr() / 2 ** 30 for a in (xs[0], xs[1], eng.b[0], eng.r[0], eng._xa[1], eng.b[1])))
    t("after construction (all zero)")
    t("  with r as b", b=eng.r[0])
    t("  with xb[1] as the correction and r[1] as the coarse b", crs=eng._xb[1], cb=eng.r[1])
    big = {"xa": xs[0], "xb": xs[1], "b": eng.b[0], "r": eng.r[0]}
    new_small = [ops.zeros(L1.ny, L1.ld) for _ in range(3)]
    small = {"b1": eng.b[1], "xa1": eng._xa[1], "xb1": eng._xb[1], "new0": new_small[0], "new2": new_small[2]}

    def one(bn, xin, xon, crsn, cbn):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        for k in range(4):
            ops.smooth_fused(big[bn], big[xin], big[xon], None, small[crsn], L0.nx, L1.nx, L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6,
                             L0.h, eng.norm, small[cbn], L1.nx, L1.t0, L1.t1)
            if k == 0:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / 3

    for cbn in ("b1", "xb1", "new0", "new2"):
        print(f"coarse b = {cbn} (GiB {small[cbn].data_ptr() / 2**30:.2f}):",
              " ".join(f"{bn},{xin}->{xon}: {one(bn, xin, xon, 'xa1', cbn):.3f}" for bn, xin, xon in
                       (("b", "xa", "xb"), ("b", "xb", "xa"), ("xa", "xb", "r"), ("r", "b", "xa"))), flush=True)
    print("correction = new0, coarse b = new2:", f"{one('b', 'xa', 'xb', 'new0', 'new2'):.3f}")
    bench.fill_rhs(eng, torch, n, 0)
    torch.cuda.synchronize()
    for a in xs:
        a.zero_()
    t("after the right-hand side was filled in")
    for a in xs:
        a.zero_()
    eng.presmooth()
    eng.cycles(4)
    t("after 4 cycles")
    t("  again, 20 launches", iters=20)
    eng.cycles(10)
    t("after 10 more cycles")


if __name__ == "__main__":
    main()
