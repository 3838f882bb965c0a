#!/usr/bin/env python3
"""Time the lexicographic SOR smoother (3 symmetric sweeps + residual, as the V-cycle calls it) on the
whole-domain level sizes, with the single-wave kernel and with the one-thread-per-row kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from multigrid_amd import _lib
    from multigrid_amd.levels import pitch
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for n, nb in [(129, 1), (65, 1), (33, 1), (17, 1), (9, 1), (33, 1024), (17, 1024)]:
        ld = pitch(n)
        b = torch.rand((nb, n, ld), dtype=torch.float64, device="cuda")
        x = torch.zeros_like(b)
        r = torch.zeros_like(b)
        for knob in (2, 0):
            _lib.check(lib.mg_set_tuning(b"sor_single_wave", knob))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(3):
                if it == 1:
                    ev0.record()
                if it >= 1:
                    for _ in range(10):
                        _lib.check(lib.mg_sor(b.data_ptr(), x.data_ptr(), r.data_ptr(), n, n, ld, 0.5, 3, 1, _lib.SOR_LEXICOGRAPHIC,
                                              nb, n * ld, st))
                else:
                    _lib.check(lib.mg_sor(b.data_ptr(), x.data_ptr(), r.data_ptr(), n, n, ld, 0.5, 3, 1, _lib.SOR_LEXICOGRAPHIC,
                                          nb, n * ld, st))
            ev1.record()
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) / 20 * 1e3
            steps = 6 * (2 * n - 4)
            print(f"{n}x{n} batch {nb} single_wave={knob}: {us:8.1f} us per call, {us / steps * 1e3:6.1f} ns per wavefront step", flush=True)
    _lib.check(lib.mg_set_tuning(b"sor_single_wave", 1))
    # grids with more rows than columns (the first whole-domain level of a weak-scaling run over rows): walked along
    # the rows (two rows per lane beyond 65 rows) or along the columns
    for ny, nx in [(129, 17), (129, 33), (65, 33), (257, 17), (513, 17), (513, 33), (1025, 17)]:
        ld = pitch(nx)
        b = torch.rand((ny, ld), dtype=torch.float64, device="cuda")
        x, r = torch.zeros_like(b), torch.zeros_like(b)
        for tr in (0, 1):
            _lib.check(lib.mg_set_tuning(b"sor_transpose", tr))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(21):
                if it == 1:
                    ev0.record()
                _lib.check(lib.mg_sor(b.data_ptr(), x.data_ptr(), r.data_ptr(), ny, nx, ld, 0.5, 3, 1, _lib.SOR_LEXICOGRAPHIC, 1, 0, st))
            ev1.record()
            torch.cuda.synchronize()
            print(f"{ny}x{nx} sor_transpose={tr}: {ev0.elapsed_time(ev1) / 20 * 1e3:8.1f} us per call", flush=True)
    _lib.check(lib.mg_set_tuning(b"sor_transpose", 1))


if __name__ == "__main__":
    main()
