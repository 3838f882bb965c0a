#!/usr/bin/env python3
"""Which of the arrays of the finest pass decides its rate?  In one process: `--count` finest-size arrays and
`--coarse` level-1-size arrays; the pass is timed with every array in turn as x_out (fixed inputs), then, with a
fast and with a slow x_out, with every array as x_in, as b, and with every coarse array as the restricted
right-hand side and as the correction."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=14)
    ap.add_argument("--coarse", type=int, default=8)
    ap.add_argument("--exp", type=int, default=14)
    ap.add_argument("--knobs-only", action="store_true")
    ap.add_argument("--data", action="store_true")
    ap.add_argument("--scan", action="store_true")
    ap.add_argument("--thirds", action="store_true", help="with --count 30: the roles x_in, x_out, b taken from the first, the "
                    "second and the third dozen of the arrays (stretches of ~96 GiB of the process's allocations)")
    args = ap.parse_args()
    import torch
    from multigrid_amd.levels import make_plan
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    n = args.exp
    N = 2 ** (n + 1) + 1
    plan = make_plan(N, N, n - 4, 1, 0, 20.0 / 2 ** (n + 1))
    L, Lc = plan.levels[0], plan.levels[1]
    norm = ops.norm_slots()
    # interleave the allocations of the two sizes, as an engine's set-up does
    A, C = [], []
    for k in range(max(args.count, args.coarse)):
        if k < args.count:
            A.append(ops.zeros(L.ny, L.ld))
        if k < args.coarse:
            C.append(ops.zeros(Lc.ny, Lc.ld))

    if args.thirds:  # one more coarse array from the end of the sequence
        C.append(ops.zeros(Lc.ny, Lc.ld))

    def t(b, xi, xo, crs, cb, reps=2):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ops.begin()
        for rep in range(reps + 1):
            ops.smooth_fused(A[b], A[xi], A[xo], None, C[crs], L.nx, Lc.nx, L.t0, L.t1, L.fix_lo, L.fix_hi, L.oj, 6, L.h, norm,
                             C[cb], Lc.nx, Lc.t0, Lc.t1)
            if rep == 0:
                ev[0].record(ops.stream)
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        return ev[0].elapsed_time(ev[1]) / reps

    def others(k, m=2):
        return [j for j in range(args.count) if j != k][:m]

    from multigrid_amd._lib import check
    for knob, vals in () if (args.data or args.scan) else (("xcd_stagger", (0, -1, 1, 3, 9)), ("fused_rows", (32, 128, 0))):
        for v in vals:
            if knob == "xcd_stagger":
                check(ops.L.mg_set_tuning(b"xcd_stagger", v))
            else:
                check(ops.L.mg_set_fused_tuning(1, v))
            out = [t(*others(k), k, 0, 1) for k in range(args.count)]
            print(f"{knob} {v:3d}: x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        check(ops.L.mg_set_tuning(b"xcd_stagger", 0))
    if args.thirds:
        c = args.count
        zone = {"A": [1, 2, 3], "B": [c // 3 + 2, c // 3 + 3, c // 3 + 4], "C": [c - 4, c - 3, c - 2]}
        print("coarse right-hand side = the first coarse array (allocated at the start); zones: A = arrays", zone["A"], "B =", zone["B"],
              "C =", zone["C"], flush=True)
        for zb in "ABC":
            for zi in "ABC":
                row = []
                for zo in "ABC":
                    pick = {}
                    used = []
                    for role, z in (("b", zb), ("xi", zi), ("xo", zo)):
                        k = next(j for j in zone[z] if j not in used)
                        used.append(k)
                        pick[role] = k
                    row.append(f"x_out {zo}: {t(pick['b'], pick['xi'], pick['xo'], 1, 0):.3f}")
                print(f"b in {zb}, x_in in {zi} | " + "  ".join(row), flush=True)
        # the coarse right-hand side from the end of the sequence instead
        last = len(C) - 1
        for zo in "ABC":
            k = zone[zo]
            print(f"coarse right-hand side allocated last, b and x_in in {zo}, x_out {zo}: {t(k[0], k[1], k[2], 1, last):.3f}", flush=True)
        return
    if args.scan:  # every array as the output, then as an input of a fast and of a slow output
        out = [t(*others(k), k, 0, 1) for k in range(args.count)]
        print("x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        return
    if args.data:
        out = [t(*others(k), k, 0, 1) for k in range(args.count)]
        print("all arrays zero      : x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        for a in A + C:
            a.normal_()
        out = [t(*others(k), k, 0, 1) for k in range(args.count)]
        print("all arrays random    : x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        for a in A + C:
            a.mul_(1e-3)
        out = [t(*others(k), k, 0, 1) for k in range(args.count)]
        print("random, 1e-3 as large: x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        for a in A + C:
            a.zero_()
        out = [t(*others(k), k, 0, 1) for k in range(args.count)]
        print("all arrays zero again: x_out = k:", " ".join(f"{v:.2f}" for v in out), flush=True)
        return
    if args.knobs_only:
        return
    out = [t(*others(k), k, 0, 1) for k in range(args.count)]
    print("x_out = k     :", " ".join(f"{v:.2f}" for v in out), flush=True)
    order = sorted(range(args.count), key=lambda k: out[k])
    fast, slow = order[0], order[-1]
    print(f"fastest output array {fast}, slowest {slow}")
    for name, xo in (("fast", fast), ("slow", slow)):
        rest = [k for k in range(args.count) if k != xo]
        r = [t(rest[0] if k != rest[0] else rest[1], k, xo, 0, 1) for k in rest]
        print(f"{name} x_out, x_in = k:", " ".join(f"{k}:{v:.2f}" for k, v in zip(rest, r)), flush=True)
        r = [t(k, rest[0] if k != rest[0] else rest[1], xo, 0, 1) for k in rest]
        print(f"{name} x_out, b = k   :", " ".join(f"{k}:{v:.2f}" for k, v in zip(rest, r)), flush=True)
        r = [t(rest[0], rest[1], xo, 0 if k != 0 else 1, k) for k in range(args.coarse)]
        print(f"{name} x_out, cb = k  :", " ".join(f"{k}:{v:.2f}" for k, v in enumerate(r)), flush=True)
        r = [t(rest[0], rest[1], xo, k, 0 if k != 0 else 1) for k in range(args.coarse)]
        print(f"{name} x_out, crs = k :", " ".join(f"{k}:{v:.2f}" for k, v in enumerate(r)), flush=True)
    # ping-pong between the two fastest, as the cycle does
    f2 = order[1]
    rest = [k for k in range(args.count) if k not in (fast, f2)]
    print(f"ping-pong {fast}<->{f2}, b = {rest[-1]}: {t(rest[-1], fast, f2, 0, 1):.2f} {t(rest[-1], f2, fast, 0, 1):.2f}")
    print("addresses (GiB):", " ".join(f"{a.data_ptr() / 2**30:.2f}" for a in A), "| coarse:", " ".join(f"{a.data_ptr() / 2**30:.2f}" for a in C))


if __name__ == "__main__":
    main()
