"""The C-ABI library loads without a GPU and exports every symbol include/mg_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "mg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared_symbols()
    for need in ("mg_jacobi", "mg_residual", "mg_restrict", "mg_prolong_add", "mg_sor", "mg_split", "mg_graph_launch"):
        assert need in names


def test_library_exports_every_declared_symbol():
    from multigrid_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in include/mg_hip.h but not exported"
    assert L.mg_abi_version() == _lib.ABI_VERSION


def test_binding_covers_the_header():
    from multigrid_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    L = _lib.lib()
    assert L.mg_last_error() is not None


def test_argument_checks_run_before_any_gpu_call():
    """Shape validation happens on the host, so it can be exercised without a GPU."""
    from multigrid_amd import _lib
    L = _lib.lib()
    assert L.mg_jacobi(None, None, None, 10, 10, 10, 1, 9, 1.0, 1.0, 1, 0, 0, None) != 0
    assert b"null" in L.mg_last_error()
    assert L.mg_jacobi(8, 16, 24, 10, 10, 4, 1, 9, 1.0, 1.0, 1, 0, 0, None) != 0
    assert b"ld" in L.mg_last_error()
    assert L.mg_restrict(8, 10, 10, 10, 16, 10, 10, 10, 1, 9, 0, 0, 1, 0, 0, None) != 0
    assert L.mg_sor(8, 16, None, 2, 2, 2, 1.0, 1, 1, 0, 1, 0, None) != 0
