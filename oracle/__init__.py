"""ctypes front end of the CPU oracle (oracle/mg_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py.  The multigrid_amd package never imports this module.

The functions mirror the reference's ``multigrid_module`` exports (core/numba_modules.py) on
C-contiguous float64 numpy arrays, plus two solver objects:

* ``BlockSolver``  -- the reference's 4-rank ``Multigrid`` (core/parallel_multigrid.py:152-348)
  run in lockstep in one process; pinned bit-for-bit by tests/golden/solve_*.npz.
* ``GlobalSolver`` -- the same V-cycle on the undecomposed grid (the layout of the GPU path).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")
_lib = None

_d = np.ctypeslib.ndpointer(dtype=np.float64, ndim=2, flags="C_CONTIGUOUS")
_dw = np.ctypeslib.ndpointer(dtype=np.float64, ndim=2, flags=("C_CONTIGUOUS", "WRITEABLE"))
_i64 = C.c_int64


def build(force=False):
    """Compile liborc.so with gcc (no-op when it is newer than its source)."""
    src = os.path.join(_HERE, "mg_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liborc.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_split.argtypes = [_d, _i64, _i64, _dw, _i64, _i64]
        L.orc_laplacian.argtypes = [_d, _dw, _i64, _i64, C.c_double]
        L.orc_residual.argtypes = [_dw, _d, _d, _i64, _i64, C.c_double]
        L.orc_smooth_sweep_jacobi.argtypes = [_d, _dw, _dw, _i64, _i64, C.c_double]
        L.orc_smooth_sweep_jacobi_damped.argtypes = [_d, _dw, _dw, _i64, _i64, C.c_double, C.c_double]
        L.orc_global_set_jacobi_omega.argtypes = [C.c_void_p, C.c_double]
        L.orc_global_set_fine_smoother.argtypes = [C.c_void_p, C.c_int]
        L.orc_smooth_sweep_rbgs.argtypes = [_d, _dw, _i64, _i64, C.c_double]
        for f in (L.orc_smooth, L.orc_smooth_altern, L.orc_smooth_altern_rb):
            f.argtypes = [_d, _dw, C.c_double, _dw, _i64, _i64, _i64]
        L.orc_coarse.argtypes = [_d, _i64, _i64, _dw, _i64, _i64, _i64, _i64]
        L.orc_interpolate_add_to.argtypes = [_d, _i64, _i64, _dw, _i64, _i64, _i64, _i64]
        L.orc_maxabs.argtypes = [C.c_void_p, _i64]
        L.orc_maxabs.restype = C.c_double
        L.orc_blocks_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_blocks_create.restype = C.c_void_p
        L.orc_blocks_destroy.argtypes = [C.c_void_p]
        L.orc_blocks_array.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(_i64)]
        L.orc_blocks_array.restype = C.POINTER(C.c_double)
        L.orc_blocks_set_verbose.argtypes = [C.c_void_p, C.c_int]
        L.orc_blocks_set_rank_parallel.argtypes = [C.c_void_p, C.c_int]
        L.orc_blocks_set_max_cycles.argtypes = [C.c_void_p, C.c_int]
        L.orc_blocks_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_blocks_solve.restype = C.c_int
        L.orc_global_create.argtypes = [_i64, _i64, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int]
        L.orc_global_create.restype = C.c_void_p
        L.orc_global_destroy.argtypes = [C.c_void_p]
        L.orc_global_set_max_cycles.argtypes = [C.c_void_p, C.c_int]
        L.orc_global_num_whole_levels.argtypes = [C.c_void_p]
        L.orc_global_num_whole_levels.restype = C.c_int
        L.orc_global_array.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]
        L.orc_global_array.restype = C.POINTER(C.c_double)
        L.orc_global_solve.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_global_solve.restype = C.c_int
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


# ---------------------------------------------------------------- multigrid_module mirror


def split(A_in, A_out, rank):
    lib().orc_split(A_in, A_in.shape[0], A_in.shape[1], A_out, A_out.shape[0], rank)


def laplacian(a, lplc, h):
    lib().orc_laplacian(a, lplc, a.shape[0], a.shape[1], h)


def residual(r, a, b, h):
    lib().orc_residual(r, a, b, a.shape[0], a.shape[1], h)


def smooth_sweep_jacobi(b, a, a_temp, h, omega=1.0):
    """omega = 1: the reference's sweep.  Otherwise a + omega * (t - a) with t the reference's value
    (an option of this build; the reference has no damping)."""
    if omega == 1.0:
        lib().orc_smooth_sweep_jacobi(b, a, a_temp, a.shape[0], a.shape[1], h)
    else:
        lib().orc_smooth_sweep_jacobi_damped(b, a, a_temp, a.shape[0], a.shape[1], h, omega)


def smooth_sweep_rbgs(b, a, h):
    """One red-black Gauss-Seidel sweep in place (an option of this build; not a reference function)."""
    lib().orc_smooth_sweep_rbgs(b, a, a.shape[0], a.shape[1], h)


def smooth(b, x, h, r, iterations):
    lib().orc_smooth(b, x, h, r, iterations, x.shape[0], x.shape[1])


def smooth_altern(b, x, h, r, iterations):
    lib().orc_smooth_altern(b, x, h, r, iterations, x.shape[0], x.shape[1])


def smooth_altern_rb(b, x, h, r, iterations):
    """Red-black ordering of smooth_altern (the GPU ordering; not a reference function)."""
    lib().orc_smooth_altern_rb(b, x, h, r, iterations, x.shape[0], x.shape[1])


def coarse(a, a_crs, ofst_i, ofst_j):
    lib().orc_coarse(a, a.shape[0], a.shape[1], a_crs, a_crs.shape[0], a_crs.shape[1], ofst_i, ofst_j)


def interpolate_add_to(a, a_new, ofst_i, ofst_j):
    lib().orc_interpolate_add_to(a, a.shape[0], a.shape[1], a_new, a_new.shape[0], a_new.shape[1], ofst_i, ofst_j)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def maxabs(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().orc_maxabs(a.ctypes.data, a.size)


def gather_blocks(blocks, M_full):
    """Stitch of core/parallel_multigrid.py:135-138 (numpy slicing, no arithmetic)."""
    nx2 = blocks[0].shape[1] - 1
    assert nx2 * 2 == M_full.shape[1] + 1
    M_full[:nx2, :nx2] = blocks[0][:-1, :-1]
    M_full[:nx2, nx2:] = blocks[1][:-1, 2:]
    M_full[nx2:, :nx2] = blocks[2][2:, :-1]
    M_full[nx2:, nx2:] = blocks[3][2:, 2:]


# ---------------------------------------------------------------- solvers


def _view(ptr, ny, nx):
    return np.ctypeslib.as_array(ptr, shape=(ny, nx))


class BlockSolver:
    """The reference's four Multigrid ranks in one process (state persists across solves)."""

    def __init__(self, n, n_para, h0, eps, rank_parallel=False):
        """rank_parallel: the four ranks' kernels run on four threads, one per rank, each kernel single-threaded
        (the reference runs as four single-threaded MPI processes); otherwise rank after rank, every kernel on
        all OpenMP threads.  The values are the same."""
        self._h = lib().orc_blocks_create(n, n_para, h0, eps)
        if not self._h:
            raise AssertionError("n_para < n required (core/parallel_multigrid.py:159)")
        lib().orc_blocks_set_rank_parallel(self._h, int(rank_parallel))
        self.n, self.n_para, self.it = n, n_para, 0
        self.x = [self.array("x_sb", k, 0) for k in range(4)]
        self.b = [self.array("b_sb", k, 0) for k in range(4)]
        self.r = [self.array("r_sb", k, 0) for k in range(4)]

    def array(self, name, rank=0, level=0):
        which = ["x_sb", "b_sb", "r_sb", "x_wl", "b_wl", "r_wl"].index(name)
        side = _i64()
        p = lib().orc_blocks_array(self._h, which, rank, level, C.byref(side))
        if not p:
            raise IndexError((name, rank, level))
        return _view(p, side.value, side.value)

    def set_max_cycles(self, m):
        """Cap on the cycles of one solve (at most the reference's 100)."""
        lib().orc_blocks_set_max_cycles(self._h, int(m))

    def solve(self):
        err = np.zeros(4)
        self.err_hist = np.zeros((4, 100))
        self.r_hist = np.zeros((4, 100))
        self.interior_hist = np.zeros(100)
        self.it = lib().orc_blocks_solve(self._h, err.ctypes.data, self.err_hist.ctypes.data, self.r_hist.ctypes.data,
                                         self.interior_hist.ctypes.data)
        self.interior_hist = self.interior_hist[: self.it]
        self.err_hist = self.err_hist[:, : self.it]
        self.r_hist = self.r_hist[:, : self.it]
        return err

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_blocks_destroy(self._h)
            self._h = None


class GlobalSolver:
    """The reference V-cycle on the undecomposed (ny x nx) grid, boundary included."""

    def __init__(self, ny, nx, n_para, h0, eps, ordering="lexicographic", fix_coarse_ascent=False, jacobi_omega=1.0,
                 fine_smoother="jacobi"):
        o = {"lexicographic": 0, "redblack": 1}[ordering]
        self._h = lib().orc_global_create(ny, nx, n_para, h0, eps, o, int(fix_coarse_ascent))
        if not self._h:
            raise ValueError("grid sides minus one must be divisible by 2**n_para, with >= 2 cells left")
        lib().orc_global_set_jacobi_omega(self._h, jacobi_omega)
        lib().orc_global_set_fine_smoother(self._h, {"jacobi": 0, "redblack": 1}[fine_smoother])
        self.n_para = n_para
        self.nwl = lib().orc_global_num_whole_levels(self._h)
        self.x, self.b, self.r = (self.array(k, 0) for k in ("x", "b", "r"))
        self.it = 0
        self._max_cycles = 100

    def array(self, name, level=0):
        which = ["x", "b", "r", "xw", "bw", "rw"].index(name)
        ny, nx = _i64(), _i64()
        p = lib().orc_global_array(self._h, which, level, C.byref(ny), C.byref(nx))
        if not p:
            raise IndexError((name, level))
        return _view(p, ny.value, nx.value)

    def set_max_cycles(self, m):
        self._max_cycles = m
        lib().orc_global_set_max_cycles(self._h, m)

    def solve(self, ncycles=0):
        err = C.c_double()
        hist = np.zeros(max(ncycles, self._max_cycles))
        self.it = lib().orc_global_solve(self._h, ncycles, C.addressof(err), hist.ctypes.data)
        self.err_hist = hist[: self.it]
        return err.value

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.orc_global_destroy(self._h)
            self._h = None
