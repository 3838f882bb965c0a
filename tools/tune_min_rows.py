#!/usr/bin/env python3
"""Cycle time against the shortest tile the fused passes may choose on small levels (knobs
"fused_min_rows", "fused_min_blocks"): the cycle graphs are re-captured for every value, several
grid sizes, interleaved rounds in one process."""
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
    lib = _lib.lib()
    for n, n_para, batch in ((14, 8, None), (13, 8, None), (11, 8, None), (7, 4, 1024)):
        N = 2 ** (n + 1) + 1
        e = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, eager_dead_levels=True, batch=batch)
        bench.fill_rhs(e, torch, n, 0)
        e.presmooth()
        configs = ((32, 2048), (8, 2048), (8, 4096), (8, 8192), (4, 2048), (4, 4096), (16, 4096), (16, 8192))
        res = {v: [] for v in configs}
        for _ in range(3):
            for v in configs:
                _lib.check(lib.mg_set_tuning(b"fused_min_rows", v[0]))
                _lib.check(lib.mg_set_tuning(b"fused_min_blocks", v[1]))
                e.close()    # drops the captured graphs: the next cycle captures them with these values
                e.cycles(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e.cycles(20)
                torch.cuda.synchronize()
                res[v].append((time.perf_counter() - t0) / 20 * 1e3)
        for v in configs:
            print(f"n={n} batch={batch} min_rows={v[0]:3d} min_blocks={v[1]:5d}: " + " ".join(f"{t:7.3f}" for t in res[v]) + " ms per cycle", flush=True)
        e.close()
        del e
        torch.cuda.empty_cache()
    _lib.check(lib.mg_set_tuning(b"fused_min_rows", 8))
    _lib.check(lib.mg_set_tuning(b"fused_min_blocks", 2048))


if __name__ == "__main__":
    main()
