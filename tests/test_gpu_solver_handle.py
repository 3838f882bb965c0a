"""The solver behind the C handle (mg_solver_*, include/mg_hip.h) against the oracle of the
reference's schedule on the global grid: per-cycle errors, x, r and every whole-domain level bit for
bit; the stop rule of solve() (tolerance, divergence latch, cycle cap); the options; error behaviour.
Marked gpu."""
import numpy as np
import pytest

import oracle
from tests.rhs import main_rhs
from tests.rhs import test_rhs as vortex_rhs

pytestmark = pytest.mark.gpu


def _rhs(gny, gnx, seed=0):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((gny, gnx))
    b[0] = b[-1] = 0
    b[:, 0] = b[:, -1] = 0
    return b


CASES = [(65, 65, 2), (129, 129, 3), (65, 129, 3), (257, 33, 2), (33, 33, 1), (1025, 1025, 4), (513, 2049, 5)]


@pytest.mark.parametrize("gny,gnx,n_para", CASES, ids=str)
@pytest.mark.parametrize("graph,recompute,ordering,fix,om", [
    (True, True, "lexicographic", False, 1.0), (False, False, "lexicographic", False, 1.0),
    (True, False, "redblack", False, 1.0), (True, True, "lexicographic", True, 0.8)], ids=str)
def test_cycles_equal_global_oracle(gny, gnx, n_para, graph, recompute, ordering, fix, om):
    from multigrid_amd.solver import NativeSolver
    b = _rhs(gny, gnx)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, ordering, fix_coarse_ascent=fix, jacobi_omega=om)
    ref.b[:] = b
    ref.solve(ncycles=5)
    big = 1 << 40
    s = NativeSolver(gny, gnx, n_para, h, use_graph=graph, coarse_ordering=ordering, fix_coarse_ascent=fix, jacobi_omega=om,
                     recompute_min_nx=0 if recompute else big, recompute_min_blocks=0 if recompute else big)
    s.upload("b", b)
    s.presmooth()
    errs = s.cycles(3) + s.cycles(2)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(s.download("x"), ref.x)
    assert np.array_equal(s.download("r"), ref.r)
    for k in range(ref.nwl):
        assert np.array_equal(s.download("xw", k), ref.array("xw", k))
        assert np.array_equal(s.download("rw", k), ref.array("rw", k))
    for l in range(1, n_para):
        assert np.array_equal(s.download("x", l), ref.array("x", l))
        if l >= 2:  # b_1 already belongs to the next cycle: the finest pass restricts its own residual
            assert np.array_equal(s.download("b", l), ref.array("b", l))
    s.close()


@pytest.mark.parametrize("n,n_para,rhs,eps", [(6, 3, "test", 0.025), (5, 2, "main", 0.025), (5, 3, "main", 1e-8), (4, 2, "test", 1e-3)])
def test_solve_stop_rule_equals_oracle(n, n_para, rhs, eps):
    """Tolerance reached, 100-cycle cap, and whatever the latch does: same count, same error, same arrays;
    a second solve() warm-starts from the first one's x as the reference's does."""
    from multigrid_amd.solver import NativeSolver
    b0, h = (vortex_rhs if rhs == "test" else main_rhs)(n)
    N = 2 ** (n + 1) + 1
    b0 = b0.copy()
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    ref = oracle.GlobalSolver(N, N, n_para, h, eps)
    ref.b[:] = b0
    s = NativeSolver(N, N, n_para, h)
    s.upload("b", b0)
    for _ in range(2):
        e_ref = ref.solve()
        e = s.solve(eps)
        assert (s.it, e) == (ref.it, e_ref)
        hist, _ = s.history()
        assert hist == ref.err_hist.tolist()
        assert np.array_equal(s.download("x"), ref.x) and np.array_equal(s.download("r"), ref.r)
    s.close()


def test_dirichlet_values_strided_host_arrays_and_device_pointers():
    import torch

    from multigrid_amd.solver import NativeSolver
    gny, gnx, n_para, h = 129, 257, 3, 0.2
    rng = np.random.default_rng(4)
    b = _rhs(gny, gnx, 4)
    x0 = np.zeros((gny, gnx))
    x0[0], x0[-1], x0[:, 0], x0[:, -1] = rng.standard_normal(gnx), rng.standard_normal(gnx), 1.5, -0.5
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0)
    ref.b[:], ref.x[:] = b, x0
    ref.solve(ncycles=3)
    s = NativeSolver(gny, gnx, n_para, h)
    wide = np.zeros((gny, gnx + 7))  # a host array with a leading dimension of its own
    wide[:, :gnx] = b
    s.upload("b", wide[:, :gnx])
    s.upload("x", x0)
    s.presmooth()
    assert s.cycles(3) == ref.err_hist.tolist()
    out = np.full((gny, gnx + 3), 9.0)
    s.download("x", 0, out[:, :gnx])
    assert np.array_equal(out[:, :gnx], ref.x) and np.all(out[:, gnx:] == 9.0)
    # device-resident access: upload from a device array, read through the device pointer
    import ctypes as C
    from multigrid_amd import _lib
    db = torch.from_numpy(b).cuda()
    _lib.check(s.L.mg_solver_upload(s._h, 1, db.data_ptr(), db.stride(0)))
    torch.cuda.synchronize()
    p, ny, nx, ld = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(s.L.mg_solver_array(s._h, 1, 0, C.byref(p), C.byref(ny), C.byref(nx), C.byref(ld)))
    assert (ny.value, nx.value) == (gny, gnx) and ld.value >= gnx
    assert np.array_equal(s.download("b"), b)
    s.close()


def test_error_behaviour():
    import ctypes as C

    from multigrid_amd import _lib
    from multigrid_amd.solver import NativeSolver
    L = _lib.lib()
    h = C.c_void_p()
    assert L.mg_solver_create(67, 65, 2, 1.0, 0, C.byref(h)) != 0 and b"divisible" in L.mg_last_error()
    assert L.mg_solver_create(65, 65, 6, 1.0, 0, C.byref(h)) != 0
    assert L.mg_solver_create(65, 65, 2, 1.0, 64, C.byref(h)) != 0 and b"flag" in L.mg_last_error()
    assert L.mg_solver_create(65, 65, 2, 0.0, 0, C.byref(h)) != 0
    assert L.mg_solver_solve(None, 0.1, None, None) != 0
    s = NativeSolver(65, 65, 2, 1.0)
    with pytest.raises(_lib.MgError):
        s.set("jacobi_omega", 1.5)
    with pytest.raises(_lib.MgError):
        s.set("no_such_option", 1)
    with pytest.raises(_lib.MgError):
        s.upload("b", np.zeros((65, 64)))
    with pytest.raises(_lib.MgError):
        s.download("r", 1)  # only the finest level keeps a residual
    with pytest.raises(_lib.MgError):
        s.download("xw", 99)
    assert L.mg_solver_upload(s._h, 2, np.zeros((65, 65)).ctypes.data, 65) != 0  # r cannot be set
    b = np.zeros((65, 65))
    b[30, 30] = np.nan
    s.upload("b", b)
    assert np.isnan(s.solve(1e-3)) and s.it == 1  # a NaN ends the loop as in the reference (err > eps is False)
    s.close()
    assert L.mg_solver_destroy(None) == 0


def test_c_program_runs_the_reference_script():
    """examples/poisson_main (plain C, no Python in the process): the reference module's script part through
    the handle API; same V-cycle counts and errors as the oracle of the global schedule."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "poisson_main")
    assert os.path.exists(exe), "build it with __graft_entry__.build() (make -C examples)"
    n, n_para = 6, 3
    out = subprocess.run([exe, str(n), str(n_para)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    b0, h = main_rhs(n)
    N = 2 ** (n + 1) + 1
    ref = oracle.GlobalSolver(N, N, n_para, h, 50 * 0.0005)
    ref.b[:] = b0
    want = []
    for _ in range(2):
        e = ref.solve()
        want.append((ref.it, f"{e / 50:.1e}"))
    got = list(zip(map(int, re.findall(r"v cycles (\d+)", out.stdout)), re.findall(r"m_err/b_max=(\S+)", out.stdout)))
    assert got == want, out.stdout
    assert float(re.search(r"residual of the downloaded solution / max\(B\): (\S+)", out.stdout).group(1)) < 1e-3


def test_tuned_placement_changes_nothing_but_where_the_arrays_lie():
    """Placement selection (the default; MG_SOLVER_NO_PLACEMENT switches it off): create times candidate allocations as
    the output of the finest pass (grids of at least 1 GiB per array) and keeps the fastest as the x arrays, zeroed;
    everything computed afterwards is the same."""
    from multigrid_amd.solver import NativeSolver
    gny, gnx, n_para = 16385, 8193, 8
    b = _rhs(gny, gnx, 2)
    out = []
    for tune in (True, False):
        s = NativeSolver(gny, gnx, n_para, 0.01, tune_placement=tune)
        assert (s.placement_candidates is not None) == tune
        s.upload("b", b)
        s.presmooth()
        errs = s.cycles(2)
        out.append((errs, s.download("x"), s.download("r")))
        s.close()
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])
    # small grids are left alone
    s = NativeSolver(257, 257, 3, 0.3, tune_placement=True)
    assert s.placement_candidates is None
    s.close()


def test_handle_cycles_run_at_the_engines_rate():
    """mg_solver_cycles reads every cycle's norm one cycle late, like SlabMultigrid.cycles: same numbers, and per
    cycle no slower than the Python engine on the same grid (8193 x 8193, the cycle takes a millisecond there, so the
    0.1 ms the host would otherwise spend waiting for each number and launching the next graph would show)."""
    import time

    import torch

    from multigrid_amd.engine import SlabMultigrid
    from multigrid_amd.solver import NativeSolver
    gny = gnx = 8193
    n_para, h, k = 8, 20.0 / 8192, 40
    b = _rhs(gny, gnx, 7)
    s = NativeSolver(gny, gnx, n_para, h)
    s.upload("b", b)
    s.presmooth()
    s.cycles(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_handle = s.cycles(k)
    t_handle = (time.perf_counter() - t0) / k
    eng = SlabMultigrid(gny, gnx, n_para, h, 0.0, world=1, rank=0, eager_dead_levels=True)
    eng.set_b_rows(torch.from_numpy(b).cuda())
    eng.presmooth()
    eng.cycles(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e_engine = eng.cycles(k)
    torch.cuda.synchronize()
    t_engine = (time.perf_counter() - t0) / k
    assert e_handle == e_engine
    print(f"handle {1e3 * t_handle:.4f} ms per cycle, engine {1e3 * t_engine:.4f} ms per cycle")
    assert t_handle <= 1.05 * t_engine, (t_handle, t_engine)
    assert np.array_equal(s.download("x"), eng.local_rows(eng.x0).cpu().numpy())
    s.close()
    eng.close()


@pytest.mark.parametrize("gny,gnx,n_para,graph,recompute", [(257, 257, 3, True, True), (1025, 513, 5, True, False), (513, 2049, 4, False, True)],
                         ids=str)
def test_dist_handle_on_one_rank_equals_oracle(gny, gnx, n_para, graph, recompute):
    """mg_dist_* with world = 1 (no communicator): the several-rank solver's layout (twelve halo rows) and schedule on
    one rank must equal the undecomposed oracle bit for bit, solve() with the stop rule included."""
    from multigrid_amd.solver import NativeDistSolver
    b = _rhs(gny, gnx, 3)
    h = 0.37
    kw = dict(recompute_min_nx=0, recompute_min_blocks=0) if recompute else {}
    s = NativeDistSolver(gny, gnx, n_para, h, None, 0, 1, use_graph=graph, **kw)
    assert s.info() == {"n_dist": 0, "first_row": 0, "owned_rows": gny, "rounds": 0}
    s.upload("b", b)
    s.presmooth()
    errs = s.cycles(3)
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0)
    ref.b[:] = b
    ref.solve(ncycles=3)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(s.download("x"), ref.x) and np.array_equal(s.download("r"), ref.r)
    # solve(): a fresh start with a tolerance; warm start afterwards
    s.upload("x", np.zeros((gny, gnx)))
    ref2 = oracle.GlobalSolver(gny, gnx, n_para, h, 0.05 * ref.err_hist[0])
    ref2.b[:] = b
    e_ref = ref2.solve()
    assert s.solve(0.05 * ref.err_hist[0]) == e_ref and s.it == ref2.it
    assert s.history()[0] == ref2.err_hist.tolist()
    assert np.array_equal(s.download("x"), ref2.x)
    s.close()
    with pytest.raises(Exception):
        NativeDistSolver(gny, gnx, n_para, h, None, 0, 2)  # several ranks need a communicator


def test_c_program_on_row_slabs():
    """examples/poisson_ranks (plain C: mg_comm_unique_id / mg_comm_init / mg_dist_*): on one rank the reference script's
    two solves with the oracle's V-cycle counts; as one rank of four rehearsed alone (a communicator of the library's own
    making, no torch in the process) the same calls run through RCCL and report their rounds per cycle."""
    import os
    import re
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "poisson_ranks")
    assert os.path.exists(exe), "build it with __graft_entry__.build() (make -C examples)"
    n, n_para = 7, 4
    out = subprocess.run([exe, str(n), str(n_para), "1", "0", "3"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    b0, h = main_rhs(n)
    N = 2 ** (n + 1) + 1
    ref = oracle.GlobalSolver(N, N, n_para, h, 50 * 0.0005)
    ref.b[:] = b0
    want = []
    for _ in range(2):
        e = ref.solve()
        want.append((ref.it, f"{e / 50:.1e}"))
    got = list(zip(map(int, re.findall(r"v cycles (\d+)", out.stdout)), re.findall(r"m_err/b_max=(\S+)", out.stdout)))
    assert got == want, out.stdout
    assert "0.0 communication rounds per cycle" in out.stdout
    rccl = "/opt/rocm/lib/librccl.so"
    if not os.path.exists(rccl):
        pytest.skip("no RCCL library at " + rccl)
    env = {k: v for k, v in os.environ.items() if k != "MG_RANKS_ID_FILE"}
    out = subprocess.run([exe, "10", "6", "4", "2", "4"], capture_output=True, text=True, timeout=150, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"rank 2 of 4 \(rows 1024\.\.1535, (\d+) distributed levels\): 4 cycles without stop test: (\S+) ms per cycle, (\S+) "
                  r"communication rounds per cycle, last max\|r\| (\S+)", out.stdout)
    assert m, out.stdout
    n_dist, ms, rounds, err = int(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(4))
    assert n_dist >= 1 and ms > 0 and np.isfinite(err)
    # per distributed level two rounds, none in front of the last one's ascent pass (one level: the exchange of x and the all-gather)
    assert rounds == (2 if n_dist == 1 else 2 * n_dist - 1)
