"""The drop-in `parallel_multigrid` surface on the GPU against the reference's own runs.

`Multigrid` (block layout) must reproduce tests/golden/solve_*.npz -- produced by the reference's
Python on four ranks -- bit for bit: iteration counts, every per-cycle per-rank error, the returned
errors and the x / r blocks of every rank, warm-started repeat solves included.  The test reads
like the reference's driver (test/test.py:33-87), one rank after the other.
"""
import glob
import os

import numpy as np
import pytest

from tests.rhs import main_rhs
from tests.rhs import test_rhs as vortex_rhs

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
SOLVES = sorted(glob.glob(os.path.join(GOLDEN, "solve_*.npz")))


@pytest.mark.parametrize("path", SOLVES, ids=lambda p: os.path.basename(p)[6:-4])
def test_block_multigrid_reproduces_reference_runs(path):
    from multigrid_amd.parallel_multigrid import BlockWorld, Multigrid, gather_blocks, split
    g = np.load(path)
    n, n_para = int(g["n"]), int(g["n_para"])
    b0, h = (vortex_rhs if str(g["rhs"]) == "test" else main_rhs)(n)
    nx1 = 2 ** n + 1
    world = BlockWorld()
    world.verbose = False
    b, a, R, solvers = [], [], [], []
    for rank in range(4):
        bk = np.zeros((nx1 + 1, nx1 + 1))
        split(b0, bk, rank)
        b.append(bk)
        a.append(np.zeros_like(bk))
        R.append(np.zeros_like(bk))
        solvers.append(Multigrid(bk, a[-1], R[-1], h, float(g["eps"]), n, n_para, rank=rank, world=world))
    its, errs, hist = [], [], []
    for _ in range(int(g["nsolves"])):
        if int(g["mutate"]):
            for rank in range(4):  # test/test.py:77-78
                b[rank][1:-1] += b[rank][0:-2]
                b[rank][:] /= 2
        errs.append([solvers[rank].solve() for rank in range(4)])
        its.append([solvers[rank].it for rank in range(4)])
        hist += world.err_history
    assert np.array_equal(np.array(its).T, g["its"])
    assert np.array_equal(np.array(errs).T, g["errs"])
    assert np.array_equal(np.array(hist).T, g["err_cycle"])
    if "x0" in g:
        for rank in range(4):
            assert np.array_equal(a[rank], g[f"x{rank}"])
            assert np.array_equal(R[rank], g[f"r{rank}"])
        for i in range(len(solvers[0].x_wl)):
            assert np.array_equal(solvers[0].x_wl[i], g[f"x_wl{i}"])
            assert np.array_equal(solvers[2].r_wl[i], g[f"r_wl{i}"])
            assert np.array_equal(solvers[1].b_wl[i], g[f"b_wl{i}"])
    else:
        for rank in range(4):
            assert np.array_equal(a[rank][::8, ::8], g[f"x{rank}_s8"])
            assert np.array_equal(R[rank][::8, ::8], g[f"r{rank}_s8"])
            assert float(np.sum(a[rank])) == float(g[f"x{rank}_sum"])
    # gather_blocks: from the four blocks, and from one rank's array through the world
    full = np.zeros_like(b0)
    gather_blocks(world, a, full)
    full2 = np.zeros_like(b0)
    gather_blocks(world, a[2], full2)
    nx2 = nx1
    assert np.array_equal(full[:nx2, :nx2], a[0][:-1, :-1]) and np.array_equal(full[nx2:, nx2:], a[3][2:, 2:])
    assert np.array_equal(full[:nx2, nx2:], a[1][:-1, 2:]) and np.array_equal(full[nx2:, :nx2], a[2][2:, :-1])
    assert np.array_equal(full, full2)


def test_rank_after_rank_calling_pattern_runs_one_solve_per_round():
    """The reference module's script part calls solve() twice per rank (core/parallel_multigrid.py:384-415).  Executed
    rank after rank -- rank 0 both calls, then rank 1 both calls, ... -- that is still two collective solves, and
    every rank gets the first round's (err, it) from its first call and the second round's from its second."""
    from multigrid_amd.parallel_multigrid import BlockWorld, Multigrid, split
    g = np.load(os.path.join(GOLDEN, "solve_m_n5_p2.npz"))
    n, n_para = int(g["n"]), int(g["n_para"])
    b0, h = main_rhs(n)
    world = BlockWorld()
    world.verbose = False
    solvers = []
    for rank in range(4):
        bk = np.zeros((2 ** n + 2, 2 ** n + 2))
        split(b0, bk, rank)
        solvers.append(Multigrid(bk, np.zeros_like(bk), np.zeros_like(bk), h, float(g["eps"]), n, n_para, rank=rank, world=world))
    rounds = 0
    orig = world.solve_all

    def counting():
        nonlocal rounds
        rounds += 1
        return orig()

    world.solve_all = counting
    errs, its = np.zeros((4, 2)), np.zeros((4, 2), dtype=int)
    for rank in range(4):
        for s in range(2):
            errs[rank, s] = solvers[rank].solve()
            its[rank, s] = solvers[rank].it
    assert rounds == 2
    assert np.array_equal(its, g["its"]) and np.array_equal(errs, g["errs"])
    # a third call of one rank starts the third round, which the others then share
    solvers[3].solve()
    solvers[0].solve()
    assert rounds == 3


def test_multigrid_argument_checks():
    from multigrid_amd.parallel_multigrid import BlockWorld, Multigrid, MgError, offsets
    w = BlockWorld()
    z = np.zeros((34, 34))
    with pytest.raises(AssertionError):
        Multigrid(z, z.copy(), z.copy(), 1.0, 0.1, 4, 2, rank=0, world=w)   # nx != 2**n + 2
    with pytest.raises(AssertionError):
        Multigrid(z, z.copy(), z.copy(), 1.0, 0.1, 5, 5, rank=0, world=w)   # n_para < n
    with pytest.raises(TypeError):
        Multigrid(z.astype(np.float32), z.copy(), z.copy(), 1.0, 0.1, 5, 2, rank=0, world=w)
    m = Multigrid(z, z.copy(), z.copy(), 1.0, 0.1, 5, 2, rank=0, world=w)
    with pytest.raises(MgError):
        m.solve()  # collective: the other three ranks are missing
    assert [len(m.x_sb), len(m.b_sb), len(m.x_wl)] == [2, 3, 4] and m.h_wl == [4.0, 8.0, 16.0, 32.0]
    assert offsets(3) == {"j": -1, "i": -1}


def test_global_multigrid_aliases_caller_arrays():
    import oracle
    from multigrid_amd.parallel_multigrid import GlobalMultigrid
    n, n_para = 6, 3
    b0, h = vortex_rhs(n)
    x, r = np.zeros_like(b0), np.zeros_like(b0)
    mg = GlobalMultigrid(b0, x, r, h, 0.025, n, n_para)
    err = mg.solve()
    ref = oracle.GlobalSolver(*b0.shape, n_para, h, 0.025)
    ref.b[:] = b0
    assert err == ref.solve() and mg.it == ref.it
    assert np.array_equal(x, ref.x) and np.array_equal(r, ref.r)
    b0 *= 0.5  # the caller mutates the right-hand side in place; the next solve sees it (warm start)
    ref.b[:] = b0
    assert mg.solve() == ref.solve() and mg.it == ref.it
    assert np.array_equal(x, ref.x)


@pytest.mark.parametrize("name", ["m_n5_p2", "m_n8_p4"])
def test_module_main_reproduces_the_reference_script(name, tmp_path, capsys):
    """`python -m multigrid_amd` == the script part of the reference module (parallel_multigrid.py:351-437):
    same V-cycle counts and errors of both solves as the recorded reference run, and the figure is written."""
    path = os.path.join(GOLDEN, f"solve_{name}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    g = np.load(path)
    assert str(g["rhs"]) == "main" and int(g["nsolves"]) == 2 and not int(g["mutate"])
    from multigrid_amd.__main__ import main
    png = tmp_path / "fig.png"
    results, a_full, R_full = main(["--n", str(int(g["n"])), "--n-para", str(int(g["n_para"])), "--png", str(png)])
    assert [it for it, _ in results] == g["its"][0].tolist()
    assert [err for _, err in results] == g["errs"].max(axis=0).tolist()
    assert png.stat().st_size > 1000
    out = capsys.readouterr().out
    assert out.count("v cycles") == 2 and f"nx = {2 ** int(g['n']) + 1}" in out
    assert np.isfinite(a_full).all() and np.abs(R_full).max() > 0


def _reference_style_driver(n, n_para, eps, nsolves, record):
    """A per-rank program written the way the reference's test driver is (test/test.py:33-87): every rank asks the
    communicator who it is, splits the global right-hand side, builds ONE Multigrid without any extra keyword, mutates
    its right-hand side and solves ten times over, reduces the error to rank 0, gathers the solution."""
    from multigrid_amd.parallel_multigrid import Multigrid, gather_blocks, residual, split
    from multigrid_amd.spmd import MPI
    comm = MPI.COMM_WORLD
    rank = comm.Get_rank()
    b0, h = vortex_rhs(n)
    nx1 = 2 ** n + 1
    b = np.zeros((nx1 + 1, nx1 + 1))
    split(b0, b, rank)
    a = np.zeros_like(b)
    R = np.zeros_like(b)
    poisson = Multigrid(b, a, R, h, eps, n, n_para)
    errs, its = [], []
    for _ in range(nsolves):
        b[1:-1] += b[0:-2]
        b /= 2
        err = poisson.solve()
        errs.append(err)
        its.append(poisson.it)
    worst = comm.reduce(errs[-1], MPI.MAX, 0)
    votes = comm.allreduce(1)
    a_full = np.zeros_like(b0)
    gather_blocks(comm, a, a_full)
    record[rank] = (errs, its, a, R, a_full, worst, votes, comm.Get_size(), poisson.n, poisson.n_para)
    return rank


def test_unchanged_per_rank_driver_runs_under_spmd_run():
    """The reference's calling pattern as it stands -- one Multigrid(b, a, R, h, eps, n, n_para) per rank, no rank= or
    world= keyword, COMM_WORLD-style reduce / allreduce, gather_blocks(comm, block, full) -- on four threads; the
    reference's own size (n = 8, n_para = 4), its recorded four-rank run as the expectation."""
    from multigrid_amd import spmd
    g = np.load(os.path.join(GOLDEN, "solve_t_n8_p4.npz"))
    n, n_para, nsolves = int(g["n"]), int(g["n_para"]), int(g["nsolves"])
    assert int(g["mutate"]) == 1 and str(g["rhs"]) == "test"
    record = {}
    world_out = spmd.run(_reference_style_driver, n, n_para, float(g["eps"]), nsolves, record)
    assert world_out == [0, 1, 2, 3]
    for rank in range(4):
        errs, its, a, R, a_full, worst, votes, size, pn, pp = record[rank]
        assert errs == g["errs"][rank].tolist() and its == g["its"][rank].tolist()
        assert np.array_equal(a[::8, ::8], g[f"x{rank}_s8"]) and np.array_equal(R[::8, ::8], g[f"r{rank}_s8"])
        assert float(np.sum(a)) == float(g[f"x{rank}_sum"])
        assert worst == (g["errs"][:, -1].max() if rank == 0 else None)
        assert votes == 4 and size == 4 and (pn, pp) == (n, n_para)
    # every rank holds the same gathered array, stitched as the reference stitches it
    nx2 = 2 ** n + 1
    full = record[0][4]
    a = [record[k][2] for k in range(4)]
    assert np.array_equal(full[:nx2, :nx2], a[0][:-1, :-1]) and np.array_equal(full[nx2:, nx2:], a[3][2:, 2:])
    assert np.array_equal(full[:nx2, nx2:], a[1][:-1, 2:]) and np.array_equal(full[nx2:, :nx2], a[2][2:, :-1])
    for rank in range(1, 4):
        assert np.array_equal(record[rank][4], full)


def test_spmd_run_hands_a_rank_error_to_the_caller():
    from multigrid_amd import spmd

    def driver():
        comm = spmd.MPI.COMM_WORLD
        if comm.Get_rank() == 2:
            raise ValueError("rank 2 gives up")
        comm.Barrier()

    with pytest.raises(ValueError, match="rank 2 gives up"):
        spmd.run(driver)
    with pytest.raises(RuntimeError):
        spmd.MPI.COMM_WORLD
