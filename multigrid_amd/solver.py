"""`NativeSolver`: the C handle API of libmg_hip.so (mg_solver_*, include/mg_hip.h) from Python.

The whole solver -- level layout, schedule, stop rule of the reference's Multigrid
(core/parallel_multigrid.py:153-348) on the undecomposed grid of one GPU -- lives in the library;
this class only moves numpy arrays in and out.  It is what a C, C++ or Fortran caller gets, and it is
used by the tests to hold that path to the same oracle as the Python engine.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MgError, check


class NativeSolver:
    def __init__(self, gny, gnx, n_para, h0, *, fix_coarse_ascent=False, coarse_ordering="lexicographic", use_graph=True,
                 eager_dead_levels=True, jacobi_omega=1.0, max_cycles=100, recompute_min_nx=None, recompute_min_blocks=None,
                 tune_placement=True):
        self.L = _lib.lib()
        flags = (_lib.SOLVER_FIX_COARSE_ASCENT if fix_coarse_ascent else 0) | \
                (_lib.SOLVER_REDBLACK if coarse_ordering == "redblack" else 0) | \
                (0 if use_graph else _lib.SOLVER_NO_GRAPH) | (0 if eager_dead_levels else _lib.SOLVER_LAZY_DEAD_LEVELS) | \
                (0 if tune_placement else _lib.SOLVER_NO_PLACEMENT)
        h = C.c_void_p()
        check(self.L.mg_solver_create(gny, gnx, n_para, h0, flags, C.byref(h)))
        self._h = h
        self.gny, self.gnx, self.n_para = gny, gnx, n_para
        self.it = 0
        if jacobi_omega != 1.0:
            self.set("jacobi_omega", jacobi_omega)
        if max_cycles != 100:
            self.set("max_cycles", max_cycles)
        if recompute_min_nx is not None:
            self.set("recompute_min_nx", recompute_min_nx)
        if recompute_min_blocks is not None:
            self.set("recompute_min_blocks", recompute_min_blocks)

    @property
    def placement_candidates(self):
        """Number of candidate arrays create timed to pick the finest level's arrays (grids of at least 1 GiB per array,
        unless tune_placement=False); None: placement not tuned."""
        g = C.c_int()
        check(self.L.mg_solver_placement(self._h, C.byref(g)))
        return None if g.value < 0 else g.value

    def set(self, key, value):
        check(self.L.mg_solver_set(self._h, key.encode(), float(value)))

    def _host(self, a):
        a = np.asarray(a)
        if a.dtype != np.float64 or a.ndim != 2 or a.strides[1] != 8 or a.strides[0] % 8:
            raise MgError("a 2-D float64 array with unit column stride is expected")
        return a

    def upload(self, name, a):
        a = self._host(a)
        if a.shape != (self.gny, self.gnx):
            raise MgError(f"expected a ({self.gny}, {self.gnx}) array")
        check(self.L.mg_solver_upload(self._h, _lib.SOLVER_ARRAYS[name], a.ctypes.data, a.strides[0] // 8))

    def shape(self, name, level=0):
        p, ny, nx, ld = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.L.mg_solver_array(self._h, _lib.SOLVER_ARRAYS[name], level, C.byref(p), C.byref(ny), C.byref(nx), C.byref(ld)))
        return ny.value, nx.value

    def download(self, name, level=0, out=None):
        ny, nx = self.shape(name, level)
        if out is None:
            out = np.empty((ny, nx))
        out = self._host(out)
        if out.shape != (ny, nx):
            raise MgError(f"expected a ({ny}, {nx}) array")
        check(self.L.mg_solver_download(self._h, _lib.SOLVER_ARRAYS[name], level, out.ctypes.data, out.strides[0] // 8))
        return out

    def presmooth(self):
        check(self.L.mg_solver_presmooth(self._h))

    def cycles(self, k):
        errs = (C.c_double * max(k, 1))()
        check(self.L.mg_solver_cycles(self._h, k, errs))
        return list(errs)[:k]

    def solve(self, eps):
        it, err = C.c_int(), C.c_double()
        check(self.L.mg_solver_solve(self._h, eps, C.byref(it), C.byref(err)))
        self.it = it.value
        return err.value

    def history(self):
        n, div = C.c_int(), C.c_int()
        check(self.L.mg_solver_history(self._h, None, 0, C.byref(n), C.byref(div)))
        errs = (C.c_double * max(n.value, 1))()
        check(self.L.mg_solver_history(self._h, errs, n.value, C.byref(n), C.byref(div)))
        return list(errs)[: n.value], bool(div.value)

    def close(self):
        if getattr(self, "_h", None):
            self.L.mg_solver_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
