#!/usr/bin/env python3
"""A/B timing inside one process (boxes differ by a few per cent): cycles(k) as is vs. per-cycle
variants of the engine.  Usage: python tools/ab_cycles.py [--exp 14]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", type=int, default=14)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    from multigrid_amd.engine import SlabMultigrid
    n = args.exp
    dummy = [torch.zeros(int(os.environ.get("MG_DUMMY_GB", "0")) << 27, dtype=torch.float64, device="cuda")]  # experiment: where the arrays land
    N = 2 ** (n + 1) + 1
    eng = SlabMultigrid(N, N, 8, 20.0 / 2 ** (n + 1), 0.025, eager_dead_levels=True,
                        tune_placement="verbose" if os.environ.get("MG_TUNE_PLACEMENT") else False)
    if os.environ.get("MG_DUMMY_FREE"):
        dummy.clear()
        torch.cuda.empty_cache()
    print("x0, b0 at GiB", round(eng.x0.data_ptr() / 2 ** 30, 2), round(eng.b0.data_ptr() / 2 ** 30, 2), "free GiB", round(torch.cuda.mem_get_info()[0] / 2 ** 30, 1), flush=True)
    bench.fill_rhs(eng, torch, n, 0)
    eng.presmooth()
    eng.cycles(2)

    def run(label, fn):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / args.steps * 1e3)
        print(f"{label:40s} " + " ".join(f"{t:7.3f}" for t in ts) + " ms per cycle", flush=True)

    def lazy_then_materialize():
        eng.ops.begin()
        for _ in range(args.steps):
            eng._run_cycle(False)
            eng._global_err()
        eng.ops.end()
        eng.materialize_residual()

    def no_residual_array():
        eng.ops.begin()
        for _ in range(args.steps):
            eng._run_cycle(False)
            eng._global_err()
        eng.ops.end()

    def store_every():
        eng.ops.begin()
        for _ in range(args.steps):
            eng._run_cycle(True)
            eng._global_err()
        eng.ops.end()

    for _ in range(2):
        run("cycles(k): last cycle stores r", lambda: eng.cycles(args.steps))
        run("k lazy cycles + residual launch", lazy_then_materialize)
        run("k lazy cycles, r never stored", no_residual_array)
        run("every cycle stores r", store_every)


if __name__ == "__main__":
    main()
