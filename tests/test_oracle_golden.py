"""The CPU oracle against the golden vectors generated from the reference's own Python.

Bit-exact throughout: the oracle restates the reference's loop and expression order and is
compiled without multiply-add fusion.  (oracle/gen_golden.py made the vectors.)
"""
import glob
import json
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
K = np.load(os.path.join(GOLDEN, "kernels.npz"))
META = json.loads(str(K["meta"]))


def cases(kernel):
    return [m for m in META if m["kernel"] == kernel]


@pytest.mark.parametrize("m", cases("residual"), ids=lambda m: f"residual{m['id']}")
def test_residual(m):
    k = m["id"]
    r = K[f"residual{k}_r0"].copy()
    oracle.residual(r, K[f"residual{k}_a"], K[f"residual{k}_b"], m["h"])
    assert np.array_equal(r, K[f"residual{k}_r"])


@pytest.mark.parametrize("m", cases("laplacian"), ids=lambda m: f"laplacian{m['id']}")
def test_laplacian(m):
    k = m["id"]
    l = K[f"laplacian{k}_l0"].copy()
    oracle.laplacian(K[f"laplacian{k}_a"], l, m["h"])
    assert np.array_equal(l, K[f"laplacian{k}_l"])


@pytest.mark.parametrize("m", cases("smooth_sweep_jacobi"), ids=lambda m: f"jacobi{m['id']}")
def test_jacobi(m):
    k = m["id"]
    a, t = K[f"jacobi{k}_a0"].copy(), K[f"jacobi{k}_t0"].copy()
    oracle.smooth_sweep_jacobi(K[f"jacobi{k}_b"], a, t, m["h"])
    assert np.array_equal(a, K[f"jacobi{k}_a"])
    assert np.array_equal(t, K[f"jacobi{k}_t"])


@pytest.mark.parametrize("m", cases("smooth") + cases("smooth_altern"), ids=lambda m: f"{m['kernel']}{m['id']}")
def test_sor(m):
    k = m["id"]
    x, r = K[f"sor{k}_x0"].copy(), K[f"sor{k}_r0"].copy()
    getattr(oracle, m["kernel"])(K[f"sor{k}_b"], x, m["h"], r, m["iterations"])
    assert np.array_equal(x, K[f"sor{k}_x"])
    assert np.array_equal(r, K[f"sor{k}_r"])


@pytest.mark.parametrize("m", cases("coarse"), ids=lambda m: f"coarse{m['id']}")
def test_coarse(m):
    k = m["id"]
    c = K[f"coarse{k}_c0"].copy()
    oracle.coarse(K[f"coarse{k}_f"], c, m["ofst_i"], m["ofst_j"])
    assert np.array_equal(c, K[f"coarse{k}_c"])


@pytest.mark.parametrize("m", cases("interpolate_add_to"), ids=lambda m: f"interp{m['id']}")
def test_interpolate(m):
    k = m["id"]
    f = K[f"interp{k}_f0"].copy()
    oracle.interpolate_add_to(K[f"interp{k}_c"], f, m["ofst_i"], m["ofst_j"])
    assert np.array_equal(f, K[f"interp{k}_f"])


@pytest.mark.parametrize("m", cases("split"), ids=lambda m: f"split{m['id']}")
def test_split(m):
    k = m["id"]
    o = np.zeros_like(K[f"split{k}_out"])
    oracle.split(K[f"split{k}_in"], o, m["rank"])
    assert np.array_equal(o, K[f"split{k}_out"])


def test_layout():
    lay = json.load(open(os.path.join(GOLDEN, "layout.json")))
    for L in lay["levels"]:
        s = oracle.BlockSolver(L["n"], L["n_para"], 1.0, 0.0)
        for i, shp in enumerate(L["x_sb"]):
            assert list(s.array("x_sb", 0, i).shape) == shp
            assert list(s.array("r_sb", 3, i).shape) == shp
        for i, shp in enumerate(L["b_sb"]):
            assert list(s.array("b_sb", 1, i).shape) == shp
        for i, shp in enumerate(L["x_wl"]):
            assert list(s.array("x_wl", 0, i).shape) == shp
        with pytest.raises(IndexError):
            s.array("x_wl", 0, len(L["x_wl"]))
    with pytest.raises(AssertionError):
        oracle.BlockSolver(4, 4, 1.0, 0.0)  # n_para < n, parallel_multigrid.py:159


SOLVES = sorted(glob.glob(os.path.join(GOLDEN, "solve_*.npz")))


def _rhs(name, n):
    from tests.rhs import main_rhs, test_rhs
    return (test_rhs if name == "test" else main_rhs)(n)


@pytest.mark.parametrize("path", SOLVES, ids=lambda p: os.path.basename(p)[6:-4])
def test_block_solver_matches_reference(path):
    """The whole 4-rank solve: iteration counts, every per-cycle error of every rank, and the
    final blocks are bit-identical to the reference run."""
    g = np.load(path)
    n, n_para = int(g["n"]), int(g["n_para"])
    b0, h = _rhs(str(g["rhs"]), n)
    assert h == float(g["h"])
    s = oracle.BlockSolver(n, n_para, h, float(g["eps"]))
    for k in range(4):
        oracle.split(b0, s.b[k], k)
    its, errs, hist, rhist = [], [], [], []
    for _ in range(int(g["nsolves"])):
        if int(g["mutate"]):
            for k in range(4):  # test/test.py:77-78
                s.b[k][1:-1] += s.b[k][0:-2]
                s.b[k][:] /= 2
        errs.append(s.solve())
        its.append(s.it)
        hist.append(s.err_hist)
        rhist.append(s.r_hist)
    assert its == g["its"][0].tolist()
    assert np.array_equal(np.array(errs).T, g["errs"])
    assert np.array_equal(np.concatenate(hist, axis=1), g["err_cycle"])
    assert np.array_equal(np.concatenate(rhist, axis=1), g["r_at_restrict"])
    if "x0" in g:
        for k in range(4):
            assert np.array_equal(s.x[k], g[f"x{k}"])
            assert np.array_equal(s.r[k], g[f"r{k}"])
        i = 0
        while f"x_wl{i}" in g:
            assert np.array_equal(s.array("x_wl", 0, i), g[f"x_wl{i}"])
            assert np.array_equal(s.array("b_wl", 0, i), g[f"b_wl{i}"])
            assert np.array_equal(s.array("r_wl", 0, i), g[f"r_wl{i}"])
            i += 1
    else:
        for k in range(4):
            assert np.array_equal(s.x[k][::8, ::8], g[f"x{k}_s8"])
            assert np.array_equal(s.r[k][::8, ::8], g[f"r{k}_s8"])


GLOBALS = sorted(glob.glob(os.path.join(GOLDEN, "global_*.npz")))


@pytest.mark.parametrize("path", GLOBALS, ids=lambda p: os.path.basename(p)[7:-4])
def test_global_solver_matches_reference_kernels_on_the_undecomposed_grid(path):
    """GlobalSolver (the oracle of the GPU slab engine) against runs in which the reference's own imported
    kernels were driven through the schedule of core/parallel_multigrid.py:264-347 on one undecomposed grid
    (oracle/gen_golden.py: run_global): per-cycle max|r|, x, r and every whole-domain array, bit for bit."""
    g = np.load(path)
    b0 = g["b0"]
    ny, nx = b0.shape
    s = oracle.GlobalSolver(ny, nx, int(g["n_para"]), float(g["h"]), 0.0, fix_coarse_ascent=bool(int(g["fix"])))
    s.b[:] = b0
    if "x_init" in g:
        s.x[:] = g["x_init"]
    s.solve(ncycles=int(g["ncycles"]))
    assert s.err_hist.tolist() == g["errs"].tolist()
    assert np.array_equal(s.x, g["x"])
    assert np.array_equal(s.r, g["r"])
    last = int(g["n_para"]) - 1
    assert np.array_equal(s.array("x", last), g["x_last"])
    assert np.array_equal(s.array("b", last), g["b_last"])
    assert len(GLOBALS) >= 9
    for i in range(s.nwl):
        for name in ("xw", "bw", "rw"):
            assert np.array_equal(s.array(name, i), g[f"{name}{i}"]), (name, i)
    assert f"xw{s.nwl}" not in g
