"""ctypes binding of libmg_hip.so (declared in include/mg_hip.h).

The library is the only compute path of this package: if it is missing or does not load, every
operation raises -- there is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmg_hip.so")
ABI_VERSION = 4
NORM_SLOTS = 64
FLAG_FINEST = 1
SOR_LEXICOGRAPHIC = 0
SOR_REDBLACK = 1
SOR_LDS_DOUBLES = 20000

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
    "mg_smooth_fused_restrict": ([_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _n, _d, _d, _p, _p, _i, _i,
                                  _i, _i, _i, _i, _i, _i, _i, _p], _n),
    "mg_set_fused_tuning": ([_n, _n], _n),
    "mg_sor": ([_p, _p, _p, _i, _i, _i, _d, _i, _n, _n, _i, _i, _p], _n),
    "mg_copy2d": ([_p, _i, _p, _i, _i, _i, _p], _n),
    "mg_block_exchange": ([_p, _i, _i, _i, _p, _p], _n),
    "mg_zero": ([_p, _i, _p], _n),
    "mg_split": ([_p, _i, _i, _i, _p, _i, _i, _i, _p], _n),
    "mg_graph_begin": ([_p], _n),
    "mg_graph_end": ([_p, C.POINTER(_p)], _n),
    "mg_graph_launch": ([_p, _p], _n),
    "mg_graph_destroy": ([_p], _n),
}


def lib():
    """Load libmg_hip.so once; raise if it is absent (build it with __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MgError(
                f"{LIB_PATH} is missing: the HIP kernels are not built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C multigrid_amd/csrc`)"
            )
        L = C.CDLL(LIB_PATH)
        for name, (args, res) in SIGNATURES.items():
            f = getattr(L, name)
            f.argtypes, f.restype = args, res
        if L.mg_abi_version() != ABI_VERSION:
            raise MgError(f"libmg_hip.so has ABI {L.mg_abi_version()}, this package expects {ABI_VERSION}")
        _lib = L
    return _lib


def check(rc):
    if rc:
        raise MgError(lib().mg_last_error().decode())
