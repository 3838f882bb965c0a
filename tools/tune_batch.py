#!/usr/bin/env python3
"""Cycle time of the batched configuration (B independent 257^2 problems) under the fused kernels' tile height and column strips per workgroup."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from multigrid_amd import _lib
    from multigrid_amd.engine import SlabMultigrid
    B, n, n_para = 1024, 7, 4
    N = 2 ** (n + 1) + 1
    lib = _lib.lib()
    for waves, rows in [(4, 0), (4, 128), (4, 256), (2, 0), (2, 128), (2, 256), (1, 0), (1, 128), (1, 256)]:
        _lib.check(lib.mg_set_fused_tuning(2, rows))
        _lib.check(lib.mg_set_tuning(b"fused_waves", waves))
        eng = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, batch=B, eager_dead_levels=True)
        bench.fill_rhs(eng, torch, n, 0)
        eng.presmooth()
        eng.cycles(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.cycles(20)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"waves {waves} rows {rows:4d}: {dt * 1e3:7.3f} ms per cycle, {B * (N - 2) ** 2 / dt / 1e9:6.1f} G unknowns*V-cycles/s", flush=True)
        eng.close()
    _lib.check(lib.mg_set_fused_tuning(2, 0))
    _lib.check(lib.mg_set_tuning(b"fused_waves", 4))


if __name__ == "__main__":
    main()
