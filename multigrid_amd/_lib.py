"""ctypes binding of libmg_hip.so (declared in include/mg_hip.h).

The library is the only compute path of this package: if it is missing or does not load, every
operation raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MG_HIP_LIB: another build of the library (A/B experiments with tools/); the default is the one beside this file
LIB_PATH = os.environ.get("MG_HIP_LIB") or os.path.join(_HERE, "libmg_hip.so")
ABI_VERSION = 8
NORM_SLOTS = 64
FLAG_FINEST = 1
SOR_LEXICOGRAPHIC = 0
SOR_REDBLACK = 1
SOR_LDS_DOUBLES = 20000
BLOCK_EXCHANGE_STATIC_MAX_SIDE = 32768

_lib = None


class MgError(RuntimeError):
    """An entry point of libmg_hip.so returned an error."""


_p = C.c_void_p
_i = C.c_int64
_d = C.c_double
_n = C.c_int

SIGNATURES = {
    "mg_abi_version": ([], _n),
    "mg_last_error": ([], C.c_char_p),
    "mg_set_tuning": ([C.c_char_p, _n], _n),
    "mg_jacobi": ([_p, _p, _p, _i, _i, _i, _i, _i, _d, _d, _i, _i, _n, _p], _n),
    "mg_jacobi_from_zero": ([_p, _p, _i, _i, _i, _i, _i, _d, _d, _i, _i, _n, _p], _n),
    "mg_residual": ([_p, _p, _p, _i, _i, _i, _i, _i, _d, _p, _i, _i, _n, _p], _n),
    "mg_laplacian": ([_p, _p, _i, _i, _i, _d, _p], _n),
    "mg_norm_reset": ([_p, _p], _n),
    "mg_norm_read": ([_p, C.POINTER(_d), _p], _n),
    "mg_maxabs": ([_p, _i, _i, _i, _p, _p], _n),
    "mg_restrict": ([_p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _n),
    "mg_prolong_add": ([_p, _i, _i, _i, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p], _n),
    "mg_smooth_fused": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _n, _n, _d, _d, _p, _i, _i, _i, _p], _n),
    "mg_smooth_fused_restrict": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _n, _d, _d, _p, _i, _p, _i, _i,
                                  _i, _i, _i, _i, _i, _i, _i, _p], _n),
    "mg_fused_also_rows": ([_i, _i, _i, _i, _i, _p], _n),
    "mg_set_fused_tuning": ([_n, _n], _n),
    "mg_sor": ([_p, _p, _p, _i, _i, _i, _d, _i, _n, _n, _i, _i, _p], _n),
    "mg_gauss_seidel_rb": ([_p, _p, _i, _i, _i, _i, _i, _d, _n, _i, _i, _p], _n),
    "mg_copy2d": ([_p, _i, _p, _i, _i, _i, _p], _n),
    "mg_block_exchange": ([_p, _i, _i, _i, _p, _p], _n),
    "mg_block_exchange_static": ([_p, _i, _i, _i, _p], _n),
    "mg_zero": ([_p, _i, _p], _n),
    "mg_split": ([_p, _i, _i, _i, _p, _i, _i, _i, _p], _n),
    "mg_graph_begin": ([_p], _n),
    "mg_graph_begin_own": ([C.POINTER(_p)], _n),
    "mg_graph_end": ([_p, C.POINTER(_p)], _n),
    "mg_graph_launch": ([_p, _p], _n),
    "mg_graph_destroy": ([_p], _n),
    "mg_comm_attach": ([_p, C.c_char_p, C.POINTER(_p)], _n),
    "mg_comm_unique_id": ([C.c_char_p, _p], _n),
    "mg_comm_init": ([_p, _n, _n, C.c_char_p, C.POINTER(_p)], _n),
    "mg_comm_info": ([_p, C.POINTER(_n), C.POINTER(_n), C.POINTER(_n)], _n),
    "mg_comm_destroy": ([_p], _n),
    "mg_halo_prepare": ([_p, _n, C.POINTER(_p), C.POINTER(_i), C.POINTER(_n), _n, C.POINTER(_p), C.POINTER(_i), C.POINTER(_n),
                         C.POINTER(_p)], _n),
    "mg_halo_exchange": ([_p, _p], _n),
    "mg_halo_destroy": ([_p], _n),
    "mg_allgather_rows": ([_p, _p, _p, _i, _p], _n),
    "mg_allreduce_max": ([_p, _p, _i, _p], _n),
    "mg_solver_create": ([_i, _i, _n, _d, _n, C.POINTER(_p)], _n),
    "mg_solver_destroy": ([_p], _n),
    "mg_solver_set": ([_p, C.c_char_p, _d], _n),
    "mg_solver_placement": ([_p, C.POINTER(_n)], _n),
    "mg_solver_upload": ([_p, _n, _p, _i], _n),
    "mg_solver_download": ([_p, _n, _n, _p, _i], _n),
    "mg_solver_array": ([_p, _n, _n, C.POINTER(_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _n),
    "mg_solver_presmooth": ([_p], _n),
    "mg_solver_cycles": ([_p, _n, C.POINTER(_d)], _n),
    "mg_solver_solve": ([_p, _d, C.POINTER(_n), C.POINTER(_d)], _n),
    "mg_solver_history": ([_p, C.POINTER(_d), _n, C.POINTER(_n), C.POINTER(_n)], _n),
    "mg_dist_create": ([_i, _i, _n, _d, _n, _p, _n, _n, _n, C.POINTER(_p)], _n),
    "mg_dist_destroy": ([_p], _n),
    "mg_dist_set": ([_p, C.c_char_p, _d], _n),
    "mg_dist_info": ([_p, C.POINTER(_n), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)], _n),
    "mg_dist_placement": ([_p, C.POINTER(_n)], _n),
    "mg_dist_upload": ([_p, _n, _p, _i], _n),
    "mg_dist_download": ([_p, _n, _p, _i], _n),
    "mg_dist_presmooth": ([_p], _n),
    "mg_dist_cycles": ([_p, _n, C.POINTER(_d)], _n),
    "mg_dist_solve": ([_p, _d, C.POINTER(_n), C.POINTER(_d)], _n),
    "mg_dist_history": ([_p, C.POINTER(_d), _n, C.POINTER(_n), C.POINTER(_n)], _n),
}
DIST_LOOPBACK = 64
SOLVER_FIX_COARSE_ASCENT, SOLVER_REDBLACK, SOLVER_NO_GRAPH, SOLVER_LAZY_DEAD_LEVELS, SOLVER_TUNE_PLACEMENT, SOLVER_NO_PLACEMENT = 1, 2, 4, 8, 16, 32
SOLVER_ARRAYS = {"x": 0, "b": 1, "r": 2, "xw": 3, "bw": 4, "rw": 5}


def lib():
    """Load libmg_hip.so once; raise if it is absent (build it with __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgError(
                f"{LIB_PATH} is missing: the HIP kernels are not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C multigrid_amd/csrc`)"
            )
        # torch ships its own HIP/HSA runtime libraries; whichever copy is loaded first serves the whole
        # process, and a second copy initialised beside it finds no GPU.  Loading torch first makes the
        # library bind to the runtime torch uses, whatever the caller imports later.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in SIGNATURES.items():
            f = getattr(L, name)
            f.argtypes, f.restype = args, res
        if L.mg_abi_version() != ABI_VERSION:
            raise MgError(f"libmg_hip.so has ABI {L.mg_abi_version()}, this package expects {ABI_VERSION}")
        _lib = L
    return _lib


# host-only units of the library: what they hold cannot change a kernel's memory traffic
_HOST_ONLY_SOURCES = ("mg_comm.hip", "mg_solver.hip", "mg_solver_dist.hip", "mg_runtime.hip")


def sources_digest():
    """sha256 over the kernel sources (csrc/*.hip and *.h except the host-only units, by name) and the compiler settings
    of the Makefile (its HIPCC / ARCH / FLAGS lines): identifies the kernel build that measurements kept under profiles/
    (HBM traffic counters) belong to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(f for f in glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h"))
                   if os.path.basename(f) not in _HOST_ONLY_SOURCES)
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(_HERE, "csrc", "Makefile")) as fh:
        for line in fh:
            if line.split("?=")[0].split(":=")[0].strip() in ("HIPCC", "ARCH", "FLAGS"):
                h.update(line.encode())
    return h.hexdigest()


def check(rc):
    if rc:
        raise MgError(lib().mg_last_error().decode())
