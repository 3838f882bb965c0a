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


class NativeDistSolver:
    """The C handle API of the solver on several GPUs (mg_dist_*, include/mg_hip.h): one object per rank (process), the
    grid cut into row slabs, halo exchanges / all-gather / norm reduction as raw RCCL calls of the library, a rank's whole
    cycle one hipGraph launch.  `comm`: an mg_comm handle (ctypes pointer) of `world` ranks -- `from_torch_group` takes the
    communicator of a torch.distributed `nccl` group.  Arrays handed in and out are the rank's owned rows of the finest
    level (`owned_rows` rows from global row `first_row`), numpy or CUDA tensors."""

    def __init__(self, gny, gnx, n_para, h0, comm, rank, world, *, dist_levels=0, fix_coarse_ascent=False,
                 coarse_ordering="lexicographic", use_graph=True, eager_dead_levels=True, jacobi_omega=1.0, max_cycles=100,
                 recompute_min_nx=None, recompute_min_blocks=None, loopback=False, tune_placement=True):
        self.L = _lib.lib()
        flags = (_lib.SOLVER_FIX_COARSE_ASCENT if fix_coarse_ascent else 0) | \
                (_lib.SOLVER_REDBLACK if coarse_ordering == "redblack" else 0) | \
                (0 if use_graph else _lib.SOLVER_NO_GRAPH) | (0 if eager_dead_levels else _lib.SOLVER_LAZY_DEAD_LEVELS) | \
                (_lib.DIST_LOOPBACK if loopback else 0) | (0 if tune_placement else _lib.SOLVER_NO_PLACEMENT)
        h = C.c_void_p()
        check(self.L.mg_dist_create(gny, gnx, n_para, h0, flags, comm, rank, world, dist_levels, C.byref(h)))
        self._h, self._own_comm = h, None
        self.gny, self.gnx, self.n_para, self.rank, self.world = gny, gnx, n_para, rank, world
        self.it = 0
        for key, val in (("jacobi_omega", jacobi_omega if jacobi_omega != 1.0 else None), ("max_cycles", max_cycles if max_cycles != 100 else None),
                         ("recompute_min_nx", recompute_min_nx), ("recompute_min_blocks", recompute_min_blocks)):
            if val is not None:
                check(self.L.mg_dist_set(self._h, key.encode(), float(val)))

    @classmethod
    def from_torch_group(cls, gny, gnx, n_para, h0, group=None, device=None, **kw):
        """Rank and size from torch.distributed, the communicator from the group's `nccl` backend."""
        import torch
        import torch.distributed as dist

        from .comm import loaded_rccl_path, torch_nccl_comm
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        t = torch.zeros(1, device=device)
        dist.all_reduce(t, group=group)  # the communicator exists once it has carried a collective
        L = _lib.lib()
        c = C.c_void_p()
        path = loaded_rccl_path()
        check(L.mg_comm_attach(C.c_void_p(torch_nccl_comm(group, device)), path.encode() if path else None, C.byref(c)))
        self = cls(gny, gnx, n_para, h0, c, dist.get_rank(group), dist.get_world_size(group), **kw)
        self._own_comm = c
        return self

    @property
    def placement_candidates(self):
        g = C.c_int()
        check(self.L.mg_dist_placement(self._h, C.byref(g)))
        return None if g.value < 0 else g.value

    def info(self):
        nd, first, own, rounds = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64()
        check(self.L.mg_dist_info(self._h, C.byref(nd), C.byref(first), C.byref(own), C.byref(rounds)))
        return {"n_dist": nd.value, "first_row": first.value, "owned_rows": own.value, "rounds": rounds.value}

    def _ptr(self, a, rows):
        if hasattr(a, "data_ptr"):  # a torch tensor, host or device
            if a.dim() != 2 or a.shape != (rows, self.gnx) or a.stride(1) != 1 or str(a.dtype) != "torch.float64":
                raise MgError(f"expected a ({rows}, {self.gnx}) float64 array with unit column stride")
            return a.data_ptr(), a.stride(0)
        a = np.asarray(a)
        if a.dtype != np.float64 or a.shape != (rows, self.gnx) or a.strides[1] != 8 or a.strides[0] % 8:
            raise MgError(f"expected a ({rows}, {self.gnx}) float64 array with unit column stride")
        return a.ctypes.data, a.strides[0] // 8

    def upload(self, name, a):
        p, ld = self._ptr(a, self.info()["owned_rows"])
        check(self.L.mg_dist_upload(self._h, _lib.SOLVER_ARRAYS[name], p, ld))

    def download(self, name, out=None):
        rows = self.info()["owned_rows"]
        if out is None:
            out = np.empty((rows, self.gnx))
        p, ld = self._ptr(out, rows)
        check(self.L.mg_dist_download(self._h, _lib.SOLVER_ARRAYS[name], p, ld))
        return out

    def presmooth(self):
        check(self.L.mg_dist_presmooth(self._h))

    def cycles(self, k):
        errs = (C.c_double * max(k, 1))()
        check(self.L.mg_dist_cycles(self._h, k, errs))
        return list(errs)[:k]

    def solve(self, eps):
        it, err = C.c_int(), C.c_double()
        check(self.L.mg_dist_solve(self._h, eps, C.byref(it), C.byref(err)))
        self.it = it.value
        return err.value

    def history(self):
        n, div = C.c_int(), C.c_int()
        check(self.L.mg_dist_history(self._h, None, 0, C.byref(n), C.byref(div)))
        errs = (C.c_double * max(n.value, 1))()
        check(self.L.mg_dist_history(self._h, errs, n.value, C.byref(n), C.byref(div)))
        return list(errs)[: n.value], bool(div.value)

    def close(self):
        if getattr(self, "_h", None):
            self.L.mg_dist_destroy(self._h)
            self._h = None
        if getattr(self, "_own_comm", None):
            self.L.mg_comm_destroy(self._own_comm)
            self._own_comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
