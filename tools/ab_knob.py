#!/usr/bin/env python3
"""A/B of a tuning knob on the V-cycle of ONE engine (same arrays): the cycle graphs are dropped and captured again
under each setting, alternating.    python tools/ab_knob.py fused_quantize 0 1 [--exp 14]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob")
    ap.add_argument("values", type=int, nargs="+")
    ap.add_argument("--exp", type=int, default=14)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--cycles", type=int, default=10)
    ap.add_argument("--also", nargs="*", default=[], help="key=value settings held fixed for the whole run")
    args = ap.parse_args()
    import torch
    import bench
    from multigrid_amd._lib import check
    from multigrid_amd.engine import SlabMultigrid
    n = args.exp
    N = 2 ** (n + 1) + 1
    eng = SlabMultigrid(N, N, max(1, n - 4), 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0, eager_dead_levels=True, tune_placement=True)
    bench.fill_rhs(eng, torch, n, 0)
    eng.presmooth()
    for kv in args.also:
        k, v = kv.split("=")
        check(eng.ops.L.mg_set_tuning(k.encode(), int(v)))
    for rnd in range(args.rounds):
        for v in args.values:
            check(eng.ops.L.mg_set_tuning(args.knob.encode(), v))
            eng.close()  # drops the captured graphs: the next cycle captures them under this setting
            eng.cycles(3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.cycles(args.cycles)
            torch.cuda.synchronize()
            print(f"round {rnd}: {args.knob} = {v}: {1e3 * (time.perf_counter() - t0) / args.cycles:.4f} ms per cycle", flush=True)


if __name__ == "__main__":
    main()
