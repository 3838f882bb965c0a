#!/usr/bin/env python3
"""Does the cycle time depend on how long the device has been busy or idle before?  One engine; the same ten cycles
timed right after set-up, after 20 s of continuous cycles, after 20 s of idling, and once more after 10 s of cycles."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from multigrid_amd.engine import SlabMultigrid
    n = 14
    N = 2 ** (n + 1) + 1
    t_start = time.perf_counter()
    eng = SlabMultigrid(N, N, n - 4, 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0, eager_dead_levels=True, tune_placement=True)
    bench.fill_rhs(eng, torch, n, 0)
    eng.presmooth()
    eng.cycles(2)

    def measure(label):
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.cycles(10)
            torch.cuda.synchronize()
            print(f"{time.perf_counter() - t_start:6.1f} s  {label}: {1e3 * (time.perf_counter() - t0) / 10:.3f} ms per cycle", flush=True)

    def busy(seconds):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            eng.cycles(20)
        torch.cuda.synchronize()

    measure("right after set-up")
    busy(20)
    measure("after 20 s of cycles")
    time.sleep(20)
    measure("after 20 s of idling")
    busy(10)
    measure("after 10 s of cycles")


if __name__ == "__main__":
    main()
