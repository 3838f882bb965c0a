#!/usr/bin/env python3
"""Finest pass of the cycle at 32769 x 32769 with x (two arrays), b and the level-1 arrays cut out of ONE
allocation, for several paddings between the arrays: does the relative position of the arrays inside
one (physically mostly contiguous) allocation decide the rate of the pass?"""
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
    plan = make_plan(N, N, 10, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    na, nc = L.ny * L.ld, Lc.ny * Lc.ld
    norm = ops.norm_slots()
    pads = [0, 1 << 10, 1 << 13, 1 << 15, 1 << 17, 1 << 19, 1 << 21, 1 << 23, 1 << 25, 3 << 23, 5 << 21, (1 << 25) + (1 << 19)]  # doubles
    for rep in range(2):
        pool = torch.zeros(3 * na + 2 * nc + 8 * max(pads) + 1024, dtype=torch.float64, device="cuda")
        for pad in pads:
            off, arrs = 0, []
            for cnt, (ny, ld) in ((na, (L.ny, L.ld)),) * 3 + ((nc, (Lc.ny, Lc.ld)),) * 2:
                arrs.append(pool[off : off + cnt].view(ny, ld))
                off = (off + cnt + pad + 15) // 16 * 16
            xa, xb, b, crs, cb = arrs
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ops.begin()
            for k in range(7):
                ops.smooth_fused(b, (xa, xb)[k & 1], (xb, xa)[k & 1], None, crs, L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6,
                                 L.h, norm, cb, Lc.nx, Lc.t0, Lc.t1)
                if k == 0:
                    ev[0].record(ops.stream)
            ev[1].record(ops.stream)
            ops.stream.synchronize()
            print(f"pool {rep} at GiB {pool.data_ptr() / 2 ** 30:.3f}, {pad * 8 / 2 ** 20:9.3f} MiB between arrays: {ev[0].elapsed_time(ev[1]) / 6:.3f} ms", flush=True)
        del pool, arrs, xa, xb, b, crs, cb
        torch.cuda.empty_cache()
        spacer = torch.empty(40 << 27, dtype=torch.float64, device="cuda")  # the second pool lands elsewhere
    del spacer


if __name__ == "__main__":
    main()
