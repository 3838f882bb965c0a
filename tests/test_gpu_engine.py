"""The slab engine on the MI355X (HIP kernels through the C ABI) against the CPU oracle and the
golden vectors of the reference.  Marked gpu.

Tolerances:
* engine vs oracle.GlobalSolver (same schedule on the undecomposed grid): bit-exact, both for the
  lexicographic coarse smoother (the reference's ordering) and for the red-black one;
* engine vs the reference's own 4-rank runs (tests/golden/solve_*.npz): the per-cycle max|r| agrees
  to 1e-4 relative on the reference's test right-hand side -- the remaining difference is the
  reference's one-exchange-old corner halo cell that enters its restriction at the point where the
  four blocks meet (the block oracle reproduces it, the global layout has no such cell);
* converged solution vs a direct solve of the 5-point system: 1e-10 relative L2.
"""
import glob
import os

import numpy as np
import pytest

import oracle
from tests.rhs import main_rhs
from tests.rhs import test_rhs as vortex_rhs

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _make(gny, gnx, n_para, h, eps, b, **kw):
    import torch
    from multigrid_amd.engine import SlabMultigrid
    eng = SlabMultigrid(gny, gnx, n_para, h, eps, world=1, rank=0, **kw)
    eng.set_b_rows(torch.from_numpy(np.ascontiguousarray(b)).cuda())
    return eng


def _rows(eng, arr):
    return eng.local_rows(arr).cpu().numpy()


def _rhs(gny, gnx, seed=0):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((gny, gnx))
    b[0] = b[-1] = 0
    b[:, 0] = b[:, -1] = 0
    return b


CASES = [(65, 65, 2), (129, 129, 3), (65, 129, 3), (257, 33, 2), (33, 33, 1), (1025, 1025, 4), (513, 2049, 5)]


@pytest.mark.parametrize("gny,gnx,n_para", CASES, ids=str)
@pytest.mark.parametrize("ordering", ["lexicographic", "redblack"])
@pytest.mark.parametrize("graph,recompute", [(True, True), (False, True), (True, False)])
def test_engine_equals_global_oracle(gny, gnx, n_para, ordering, graph, recompute):
    b = _rhs(gny, gnx)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, ordering)
    ref.b[:] = b
    ref.solve(ncycles=4)
    # recompute: the coarse levels' post-smoothing passes recompute the pre-smoothing (forced here: by
    # default only large levels do)
    eng = _make(gny, gnx, n_para, h, 0.0, b, coarse_ordering=ordering, use_graph=graph, recompute_presmoothing=recompute,
                recompute_min_nx=0, recompute_min_blocks=0)
    eng.presmooth()
    errs = eng.cycles(4)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)
    assert np.array_equal(_rows(eng, eng.r0), ref.r)
    eng.sync_whole_levels()
    for k, W in enumerate(eng.plan.whole):
        assert np.array_equal(eng.xw[k].cpu().numpy()[:, : W.nx], ref.array("xw", k))
    eng.close()


@pytest.mark.parametrize("gny,gnx,n_para", [(129, 129, 3), (1025, 1025, 4), (513, 2049, 5)], ids=str)
@pytest.mark.parametrize("graph", [True, False])
def test_dead_levels_every_cycle_on_the_side_stream(gny, gnx, n_para, graph):
    """eager_dead_levels: the whole-domain levels below the first run every cycle, beside the ascent on a
    second stream (inside the captured graph too); every array equals the oracle's."""
    b = _rhs(gny, gnx, 5)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, "lexicographic")
    ref.b[:] = b
    ref.solve(ncycles=3)
    eng = _make(gny, gnx, n_para, h, 0.0, b, use_graph=graph, eager_dead_levels=True)
    eng.presmooth()
    assert eng.cycles(3) == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)
    assert np.array_equal(_rows(eng, eng.r0), ref.r)
    eng.ops.synchronize()
    for k, W in enumerate(eng.plan.whole):
        assert np.array_equal(eng.xw[k].cpu().numpy()[:, : W.nx], ref.array("xw", k))
    eng.close()


@pytest.mark.parametrize("B,N,n_para", [(5, 65, 2), (6, 257, 4)], ids=str)
@pytest.mark.parametrize("graph", [True, False])
def test_batch_of_problems_equals_independent_solves(B, N, n_para, graph):
    """batch=B (the 1024 x 257^2 configuration in small): every launch covers all problems; each problem
    gets the result of its own undecomposed solve, bit for bit; the norm is the max over the batch."""
    import torch
    from multigrid_amd.engine import SlabMultigrid
    rng = np.random.default_rng(9)
    bs = rng.standard_normal((B, N, N))
    bs[:, 0] = bs[:, -1] = 0
    bs[:, :, 0] = bs[:, :, -1] = 0
    h = 0.41
    refs = []
    for k in range(B):
        ref = oracle.GlobalSolver(N, N, n_para, h, 0.0)
        ref.b[:] = bs[k]
        ref.solve(ncycles=3)
        refs.append(ref)
    eng = SlabMultigrid(N, N, n_para, h, 0.0, world=1, rank=0, batch=B, use_graph=graph, eager_dead_levels=True,
                        recompute_min_nx=0 if graph else 1024, recompute_min_blocks=0 if graph else 1024)
    eng.set_b_rows(torch.from_numpy(bs).cuda())
    eng.presmooth()
    errs = eng.cycles(3)
    assert errs == [max(r.err_hist[c] for r in refs) for c in range(3)]
    x, r = _rows(eng, eng.x0), _rows(eng, eng.r0)
    eng.ops.synchronize()
    for k in range(B):
        assert np.array_equal(x[k], refs[k].x)
        assert np.array_equal(r[k], refs[k].r)
        for lv, W in enumerate(eng.plan.whole):
            assert np.array_equal(eng.xw[lv][k].cpu().numpy()[:, : W.nx], refs[k].array("xw", lv))
    eng.close()


@pytest.mark.parametrize("graph", [True, False])
def test_batch_stops_every_problem_on_its_own(graph):
    """batch_stop='each': right-hand sides of different scale in one batch; every problem stops at the cycle at which
    its own solve() stops and ends with that solve's x, r, cycle count and max|r| (independent oracle runs), also on a
    second, warm-started solve; one problem carries a NaN and stops at once with a NaN."""
    import torch
    from multigrid_amd.engine import SlabMultigrid
    B, N, n_para, h, eps = 7, 129, 3, 20.0 / 128, 0.025
    b0, _ = vortex_rhs(6)
    b0 = b0.copy()
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    scales = [1.0, 1e-3, 30.0, 0.2, 4.0, 1e-6, 1.0]
    bs = np.stack([sc * np.roll(b0, 5 * k, axis=1) for k, sc in enumerate(scales)])
    bs[:, :, 0] = bs[:, :, -1] = 0
    bs[6, 40, 40] = np.nan
    refs = []
    for k in range(B):
        ref = oracle.GlobalSolver(N, N, n_para, h, eps)
        ref.b[:] = bs[k]
        refs.append(ref)
    eng = SlabMultigrid(N, N, n_para, h, eps, world=1, rank=0, batch=B, batch_stop="each", use_graph=graph)
    eng.set_b_rows(torch.from_numpy(bs).cuda())
    for rnd in range(2):
        if rnd == 1:  # the reference driver's change of the right-hand side between solves (test/test.py:77-78); x stays
            for ref in refs:
                ref.b[1:-1] += ref.b[0:-2].copy()
                ref.b[:] /= 2
            eng.set_b_rows(torch.from_numpy(np.stack([ref.b for ref in refs])).cuda())
        want = [ref.solve() for ref in refs]
        err = eng.solve()
        its = [ref.it for ref in refs]
        assert eng.problem_its.tolist() == its, (rnd, eng.problem_its, its)
        assert len(set(its)) >= (4 if rnd == 0 else 2)  # the batch does hold problems that stop at different cycles
        for k in range(B - 1):
            assert eng.problem_errs[k] == want[k], (rnd, k)
            assert np.array_equal(_rows(eng, eng.x0)[k], refs[k].x), (rnd, k)
            assert np.array_equal(_rows(eng, eng.r0)[k], refs[k].r), (rnd, k)
        assert np.isnan(eng.problem_errs[6]) and np.isnan(want[6]) and eng.problem_its[6] == 1 and np.isnan(err)
        assert eng.it == max(its)
    # the default rule looks at the maximum over the batch: all problems run as long as the slowest
    eng2 = SlabMultigrid(N, N, n_para, h, eps, world=1, rank=0, batch=B - 1, use_graph=graph)
    eng2.set_b_rows(torch.from_numpy(bs[: B - 1]).cuda())
    eng2.solve()
    assert eng2.it == max(ref.it for ref in [oracle_first_solve(bs[k], N, n_para, h, eps) for k in range(B - 1)])
    eng.close()
    eng2.close()


def oracle_first_solve(b, N, n_para, h, eps):
    ref = oracle.GlobalSolver(N, N, n_para, h, eps)
    ref.b[:] = b
    ref.solve()
    return ref


def test_fix_coarse_ascent_equals_oracle():
    gny = gnx = 257
    b = _rhs(gny, gnx, 3)
    ref = oracle.GlobalSolver(gny, gnx, 3, 0.5, 0.0, "lexicographic", fix_coarse_ascent=True)
    ref.b[:] = b
    ref.solve(ncycles=3)
    eng = _make(gny, gnx, 3, 0.5, 0.0, b, fix_coarse_ascent=True, keep_ascent_residuals=True)
    eng.presmooth()
    assert eng.cycles(3) == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)


def _global_rhs_from_blocks(g, nsolve_index):
    """The right-hand sides of solves 0..nsolve_index of a golden run, each as (global grid, the four
    blocks).  test.py mutates the four blocks in place before every solve (test/test.py:77-78) -- from
    the second mutation on the copies of a shared line in two blocks no longer agree, because a
    block's first row is never updated -- and the global grid is their gather_blocks stitch."""
    n = int(g["n"])
    b0, h = (vortex_rhs if str(g["rhs"]) == "test" else main_rhs)(n)
    blocks = [np.zeros((2 ** n + 2, 2 ** n + 2)) for _ in range(4)]
    for k in range(4):
        oracle.split(b0, blocks[k], k)
    out = []
    for _ in range(nsolve_index + 1):
        if int(g["mutate"]):
            for k in range(4):
                blocks[k][1:-1] += blocks[k][0:-2]
                blocks[k][:] /= 2
        full = np.zeros_like(b0)
        oracle.gather_blocks(blocks, full)
        out.append((full, [blk.copy() for blk in blocks]))
    return out, h


# (golden run, tolerance on the per-cycle max|r|, first cycle from which the tolerance and the 5 % criterion on
# the reduction per V-cycle are asserted; cycles count from 1).  The engine is bit-identical to the undecomposed
# schedule; the reference's 4-block run deviates from that by its one-exchange-old corner halo cells at the point
# where the four blocks meet (the diagonal neighbour's cell reaches a block only with the second exchange after
# it changed: weight 1/16 in the restriction, 1/4 in the prolongation, of one cell per level), which matters more
# the fewer cells a block has on its coarsest Jacobi level (deep n_para on a small grid) and when the source sits
# on the junction.
# The "m_" runs are the reference module's own script (core/parallel_multigrid.py:351-415): a single Gaussian
# centred exactly on that junction, where the residual has its maximum.  There the reference's first cycles are
# dominated by the artefact -- its own per-rank errors (which include the stale halo copies) read 26.6, 20.5, 1.6
# at n = 5 while the residual on the block interiors is 2.2, 0.16, 0.10 -- and the undecomposed schedule differs:
#   max|r| after cycles 1, 2, 3 relative to the reference's interior max|r|
#       m_n5_p2: -10 %, +116 %, +36 %     m_n6_p3: -14 %, +27 %, +30 %     m_n8_p4: -22 %, +51 %, +23 %
#   reduction of cycles 2, 3, 4 relative to the reference's
#       m_n5_p2: +142 %, -37 %, -23 %     m_n6_p3: +47 %, +2.5 %, -20 %    m_n8_p4: +93 %, -19 %, -14 %
# From cycle 4 on max|r| is within 6 %, from cycle 5 on within 1.6 % and falling (0.1 % at the last cycle of
# m_n8_p4); the reduction of cycle 5 is within 4.4 % and that of every later cycle within 0.4 %.  So the 5 %
# criterion on the reduction per V-cycle holds from cycle 5 on and is asserted from there; cycles 2-4 of these
# three runs are the ones the slab layout misses (the block-layout Multigrid reproduces them bit for bit,
# tests/test_gpu_dropin.py).  Both runs reach the reference's tolerance in the same number of cycles.
HISTORY_CASES = [("t_n8_p4", 1e-4, 1), ("t_n7_p4", 1e-4, 1), ("t_n6_p3", 1e-4, 1), ("t_n6_p5", 5e-3, 1), ("t_n5_p4", 1e-2, 1),
                 ("t_n5_p2", 1e-3, 1), ("t_n4_p3", 1e-2, 1), ("m_n5_p2", 0.06, 4), ("m_n6_p3", 0.06, 4), ("m_n8_p4", 0.06, 4)]
# the deviations of the first cycles listed above, as bounds: (max|r| of cycles 1-3, reduction of cycles 2-4)
M_EARLY = {"m_n5_p2": (1.2, 1.5), "m_n6_p3": (0.35, 0.5), "m_n8_p4": (0.55, 1.0)}


@pytest.mark.parametrize("name,tol,first", HISTORY_CASES, ids=lambda v: str(v))
def test_history_matches_reference_runs(name, tol, first):
    """Same right-hand sides as the reference's 4-rank runs (warm-started solves included).  The
    reference's own history is reproduced by the block oracle (bit-exact, test_oracle_golden.py);
    its per-rank error includes stale halo copies of the neighbours' residual, so the comparison
    uses the block oracle's max|r| over the block interiors, i.e. the same quantity the engine
    reports: the max-norm of the residual over the whole grid."""
    import torch
    g = np.load(os.path.join(GOLDEN, f"solve_{name}.npz"))
    n, n_para = int(g["n"]), int(g["n_para"])
    N = 2 ** (n + 1) + 1
    rhs, h = _global_rhs_from_blocks(g, int(g["nsolves"]) - 1)
    blk = oracle.BlockSolver(n, n_para, h, float(g["eps"]))
    eng = _make(N, N, n_para, h, float(g["eps"]), rhs[0][0])
    pos = 0
    for s in range(int(g["nsolves"])):
        for k in range(4):
            blk.b[k][:] = rhs[s][1][k]
        blk.solve()
        it_ref = int(g["its"][0][s])
        assert blk.it == it_ref and np.array_equal(blk.err_hist, g["err_cycle"][:, pos : pos + it_ref])
        pos += it_ref
        eng.set_b_rows(torch.from_numpy(rhs[s][0]).cuda())
        eng.presmooth()
        hist = np.array(eng.cycles(it_ref))
        rel = np.abs(hist - blk.interior_hist) / blk.interior_hist
        ratio = (hist[1:] / hist[:-1]) / (blk.interior_hist[1:] / blk.interior_hist[:-1])  # ratio[k]: reduction of cycle k + 2
        xg = np.zeros((N, N))
        oracle.gather_blocks(blk.x, xg)
        xrel = np.linalg.norm(_rows(eng, eng.x0) - xg) / np.linalg.norm(xg)
        # a warm-started solve of an "m_" run starts from the x of the first one: all of its cycles count as late
        f = first if s == 0 else 1
        assert rel[f - 1 :].max() < tol, (s, rel)
        assert rel[-1] < min(tol, 0.025), (s, rel)
        # reduction per V-cycle: the 5 % criterion
        assert np.all(np.abs(ratio[max(f - 1, 0) :] - 1) < 0.05), (s, ratio)
        assert np.all(np.abs(ratio[f + 1 :] - 1) < 0.005), (s, ratio)
        if name in M_EARLY:
            assert xrel < 1e-3, (s, xrel)
            if s == 0:  # the documented misses of the first cycles stay what the table above says
                assert rel[: f - 1].max() < M_EARLY[name][0] and np.abs(ratio[: f - 1] - 1).max() < M_EARLY[name][1], (rel, ratio)
                # both schedules reach the tolerance of the run in the same cycle
                assert (hist > float(g["eps"])).sum() == (blk.interior_hist > float(g["eps"])).sum()
        else:
            assert xrel < max(tol, 1e-4) * 0.1, (s, xrel)


GLOBAL_RUNS = sorted(glob.glob(os.path.join(GOLDEN, "global_*.npz")))


@pytest.mark.parametrize("path", GLOBAL_RUNS, ids=lambda p: os.path.basename(p)[7:-4])
@pytest.mark.parametrize("graph,recompute", [(True, True), (False, False)])
def test_engine_equals_reference_kernels_on_the_undecomposed_grid(path, graph, recompute):
    """The engine against runs of the reference's own kernels (imported from the reference checkout when the
    fixtures were made, oracle/gen_golden.py: run_global) driven through the schedule of
    core/parallel_multigrid.py:264-347 on the undecomposed grid: per-cycle max|r|, x, r and the whole-domain
    arrays, bit for bit -- no oracle in between."""
    import torch
    g = np.load(path)
    b0 = g["b0"]
    gny, gnx = b0.shape
    fix = bool(int(g["fix"]))
    eng = _make(gny, gnx, int(g["n_para"]), float(g["h"]), 0.0, b0, use_graph=graph, fix_coarse_ascent=fix,
                recompute_presmoothing=recompute, recompute_min_nx=0, recompute_min_blocks=0, eager_dead_levels=True)
    if "x_init" in g:
        eng.set_x_rows(torch.from_numpy(g["x_init"]).cuda())
    eng.presmooth()
    assert eng.cycles(int(g["ncycles"])) == g["errs"].tolist()
    assert np.array_equal(_rows(eng, eng.x0), g["x"])
    assert np.array_equal(_rows(eng, eng.r0), g["r"])
    eng.ops.synchronize()
    for k, W in enumerate(eng.plan.whole):
        assert np.array_equal(eng.xw[k].cpu().numpy()[:, : W.nx], g[f"xw{k}"]), k
    eng.close()


def test_solve_counts_cycles_like_the_reference():
    """solve() (stop rule included) on the reference's own test size, n = 8, n_para = 4: 22 V-cycles
    for the first solve and 3 for the warm-started second one, as the reference's run."""
    import torch
    g = np.load(os.path.join(GOLDEN, "solve_t_n8_p4.npz"))
    rhs, h = _global_rhs_from_blocks(g, 1)
    eng = _make(513, 513, 4, h, 0.025, rhs[0][0])
    errs = []
    for s in range(2):
        eng.set_b_rows(torch.from_numpy(rhs[s][0]).cuda())
        errs.append(eng.solve())
        assert eng.it == int(g["its"][0][s])
        assert abs(errs[-1] - g["errs"][:, s].max()) / errs[-1] < 1e-4


def test_converged_solution_matches_direct_solve():
    """Run V-cycles to stagnation and compare with a sparse direct solve of the same 5-point system."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, n_para = 5, 2
    b0, h = main_rhs(n)
    N = 2 ** (n + 1) + 1
    m = N - 2
    T = sp.diags([np.ones(m - 1), -2 * np.ones(m), np.ones(m - 1)], [-1, 0, 1])
    A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))) / h ** 2
    x_direct = np.zeros((N, N))
    x_direct[1:-1, 1:-1] = spla.spsolve(A.tocsc(), b0[1:-1, 1:-1].ravel()).reshape(m, m)
    for fix in (False, True):
        eng = _make(N, N, n_para, h, 0.0, b0, fix_coarse_ascent=fix)
        eng.presmooth()
        err, total = 1.0, 0
        while err > 1e-11 and total < 40000:
            err = eng.cycles(500)[-1]
            total += 500
        x = _rows(eng, eng.x0)
        rel = np.linalg.norm(x - x_direct) / np.linalg.norm(x_direct)
        assert rel < 1e-10, (fix, total, err, rel)
        eng.close()


def test_converged_solution_matches_direct_solve_at_513():
    """The reference's own size (513 x 513, n = 8, n_para = 4, its test driver's right-hand side): with the textbook
    ascent through the whole-domain levels and damped sweeps (the reference's undamped schedule stalls on the
    checkerboard mode long before) the cycle converges to rounding level, and the solution equals a sparse direct
    solve of the same 5-point system to 1e-10 relative L2."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, n_para = 8, 4
    b0, h = vortex_rhs(n)
    b0 = b0.copy()
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    N = 2 ** (n + 1) + 1
    m = N - 2
    T = sp.diags([np.ones(m - 1), -2 * np.ones(m), np.ones(m - 1)], [-1, 0, 1])
    A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))) / h ** 2
    x_direct = np.zeros((N, N))
    x_direct[1:-1, 1:-1] = spla.spsolve(A.tocsc(), b0[1:-1, 1:-1].ravel()).reshape(m, m)
    eng = _make(N, N, n_para, h, 0.0, b0, fix_coarse_ascent=True, jacobi_omega=0.8)
    eng.presmooth()
    errs = eng.cycles(60)
    x = _rows(eng, eng.x0)
    rel = np.linalg.norm(x - x_direct) / np.linalg.norm(x_direct)
    assert rel < 1e-10, (errs[-1], rel)
    assert errs[-1] < 1e-8 * errs[0]
    eng.close()


@pytest.mark.parametrize("gny,gnx,n_para", [(129, 129, 3), (1025, 1025, 4), (513, 2049, 5)], ids=str)
@pytest.mark.parametrize("graph,recompute,fused", [(True, True, True), (False, False, True), (True, False, False)])
def test_damped_jacobi_equals_oracle(gny, gnx, n_para, graph, recompute, fused):
    """jacobi_omega = 0.8 (every sweep x + omega (t - x); an option of this build, the reference's omega
    is 1): bit-identical to the oracle's damped schedule, fused or not, recomputed pre-smoothing or not."""
    b = _rhs(gny, gnx, 2)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, fix_coarse_ascent=True, jacobi_omega=0.8)
    ref.b[:] = b
    ref.solve(ncycles=4)
    eng = _make(gny, gnx, n_para, h, 0.0, b, fix_coarse_ascent=True, jacobi_omega=0.8, use_graph=graph, fused=fused,
                recompute_presmoothing=recompute, recompute_min_nx=0, recompute_min_blocks=0)
    eng.presmooth()
    errs = eng.cycles(4)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)
    assert np.array_equal(_rows(eng, eng.r0), ref.r)
    eng.close()


@pytest.mark.parametrize("gny,gnx,n_para", [(129, 129, 3), (1025, 1025, 4), (513, 2049, 5), (65, 33, 1)], ids=str)
@pytest.mark.parametrize("graph,fix", [(True, False), (False, True)])
def test_redblack_gauss_seidel_on_the_fine_levels_equals_oracle(gny, gnx, n_para, graph, fix):
    """fine_smoother='redblack' (red-black Gauss-Seidel sweeps in place of the Jacobi sweeps of the fine levels; an
    option of this build): bit-identical to the oracle's restatement of that schedule."""
    b = _rhs(gny, gnx, 6)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, fix_coarse_ascent=fix, fine_smoother="redblack")
    ref.b[:] = b
    ref.solve(ncycles=4)
    eng = _make(gny, gnx, n_para, h, 0.0, b, fix_coarse_ascent=fix, fine_smoother="redblack", use_graph=graph)
    assert not eng.fused
    eng.presmooth()
    assert eng.cycles(4) == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)
    assert np.array_equal(_rows(eng, eng.r0), ref.r)
    eng.close()


def test_redblack_textbook_cycle_converges_to_rounding_level():
    """Red-black Gauss-Seidel on the fine levels with the textbook ascent: like the damped Jacobi cycle it reaches a
    tolerance eleven orders below the right-hand side, in fewer cycles (its smoothing factor is 0.25 against 0.6)."""
    n, n_para = 6, 3
    b0, h = vortex_rhs(n)
    N = 2 ** (n + 1) + 1
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    eng = _make(N, N, n_para, h, 50 * 1e-11, b0, fix_coarse_ascent=True, fine_smoother="redblack")
    err = eng.solve()
    assert err <= 50 * 1e-11 and eng.it <= 14 and not eng.diverged, (err, eng.it)
    damped = _make(N, N, n_para, h, 50 * 1e-11, b0, fix_coarse_ascent=True, jacobi_omega=0.8)
    damped.solve()
    assert eng.it < damped.it
    rel = np.linalg.norm(_rows(eng, eng.x0) - _rows(damped, damped.x0)) / np.linalg.norm(_rows(eng, eng.x0))
    assert rel < 1e-10, rel
    eng.close()
    damped.close()


def test_damped_textbook_cycle_converges_to_rounding_level():
    """With damping and the textbook ascent the stop rule is met at a tolerance eleven orders below the
    right-hand side within 25 cycles, and the solution equals a sparse direct solve (the reference's
    own schedule needs tens of thousands of cycles for that, see above)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    n, n_para = 6, 3
    b0, h = vortex_rhs(n)
    N = 2 ** (n + 1) + 1
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    m = N - 2
    T = sp.diags([np.ones(m - 1), -2 * np.ones(m), np.ones(m - 1)], [-1, 0, 1])
    A = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m))) / h ** 2
    x_direct = np.zeros((N, N))
    x_direct[1:-1, 1:-1] = spla.spsolve(A.tocsc(), b0[1:-1, 1:-1].ravel()).reshape(m, m)
    eng = _make(N, N, n_para, h, 50 * 1e-11, b0, fix_coarse_ascent=True, jacobi_omega=0.8)
    err = eng.solve()
    assert err <= 50 * 1e-11 and eng.it <= 25 and not eng.diverged, (err, eng.it)
    rel = np.linalg.norm(_rows(eng, eng.x0) - x_direct) / np.linalg.norm(x_direct)
    assert rel < 1e-10, rel
    eng.close()


def test_rectangular_and_nan():
    """Rectangular grids work; a NaN in the right-hand side surfaces in the returned norm."""
    b = _rhs(129, 513, 5)
    eng = _make(129, 513, 3, 0.1, 1e-3, b)
    ref = oracle.GlobalSolver(129, 513, 3, 0.1, 1e-3)
    ref.b[:] = b
    assert eng.solve() == ref.solve()
    assert eng.it == ref.it
    b[40, 40] = np.nan
    eng2 = _make(129, 513, 3, 0.1, 1e-3, b)
    assert np.isnan(eng2.solve())


@pytest.mark.parametrize("fused", [True, False])
def test_nonzero_dirichlet_boundary(fused):
    """Boundary values of x are data: they enter the finest-level sweeps and residual and are never
    changed (the corrections of the coarser levels have zero boundaries).  Engine == oracle bit for
    bit, and the converged solution solves the Dirichlet problem with those boundary values."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    import torch
    gny, gnx, n_para, h = 65, 129, 3, 0.05
    rng = np.random.default_rng(12)
    b = _rhs(gny, gnx, 12)
    x0 = np.zeros((gny, gnx))
    x0[0], x0[-1], x0[:, 0], x0[:, -1] = (rng.standard_normal(gnx), rng.standard_normal(gnx), rng.standard_normal(gny),
                                           rng.standard_normal(gny))
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0)
    ref.b[:] = b
    ref.x[:] = x0
    ref.solve(ncycles=4)
    eng = _make(gny, gnx, n_para, h, 0.0, b, fused=fused)
    eng.set_x_rows(torch.from_numpy(x0).cuda())
    eng.presmooth()
    assert eng.cycles(4) == ref.err_hist.tolist()
    assert np.array_equal(_rows(eng, eng.x0), ref.x)
    assert np.array_equal(_rows(eng, eng.r0), ref.r)
    # direct solve: A x_int = b_int - (boundary contributions) / h^2
    my, mx = gny - 2, gnx - 2
    Ty = sp.diags([np.ones(my - 1), -2 * np.ones(my), np.ones(my - 1)], [-1, 0, 1])
    Tx = sp.diags([np.ones(mx - 1), -2 * np.ones(mx), np.ones(mx - 1)], [-1, 0, 1])
    A = (sp.kron(Ty, sp.identity(mx)) + sp.kron(sp.identity(my), Tx)) / h ** 2
    rhs = b[1:-1, 1:-1].copy()
    rhs[0] -= x0[0, 1:-1] / h ** 2
    rhs[-1] -= x0[-1, 1:-1] / h ** 2
    rhs[:, 0] -= x0[1:-1, 0] / h ** 2
    rhs[:, -1] -= x0[1:-1, -1] / h ** 2
    x_direct = x0.copy()
    x_direct[1:-1, 1:-1] = spla.spsolve(A.tocsc(), rhs.ravel()).reshape(my, mx)
    err, total = 1.0, 0
    while err > 1e-9 and total < 40000:
        err = eng.cycles(500)[-1]
        total += 500
    x = _rows(eng, eng.x0)
    rel = np.linalg.norm(x - x_direct) / np.linalg.norm(x_direct)
    assert rel < 1e-10, (total, err, rel)
    eng.close()


# several knobs at once, each away from its default: (key, value, default)
KNOB_SETS = [[("stream_min_mib", 0, 256), ("fused_lockstep", 3, 1), ("fused_rows", 24, 0)],
             [("stream_min_mib", 0, 256), ("fused_waves", 2, 4), ("fused_rows_shallow", 8, 32), ("fused_rows_deep", 16, 128)],
             [("fused_lockstep", 0, 1), ("xcd_stagger", -1, 0), ("fused_min_rows", 4, 8), ("fused_min_blocks", 1 << 20, 2048)]]


@pytest.mark.parametrize("knobs", KNOB_SETS, ids=lambda ks: "+".join(f"{k}={v}" for k, v, _ in ks))
@pytest.mark.parametrize("fused", [True, False])
def test_cycle_equals_oracle_under_tuning_knobs(knobs, fused):
    """Whole V-cycles with the tuning knobs away from their defaults (streaming stores on every array, other tile
    heights, strips per workgroup and barrier spacings): still the oracle's numbers, bit for bit."""
    from multigrid_amd import _lib as L
    gny, gnx, n_para, h = 513, 2049, 5, 0.37
    b = _rhs(gny, gnx, 11)
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0)
    ref.b[:] = b
    ref.solve(ncycles=3)
    lib = L.lib()
    for key, value, _ in knobs:
        L.check(lib.mg_set_tuning(key.encode(), value))
    try:
        eng = _make(gny, gnx, n_para, h, 0.0, b, fused=fused, recompute_min_nx=0, recompute_min_blocks=0, eager_dead_levels=True)
        eng.presmooth()
        errs = eng.cycles(3)
        assert errs == ref.err_hist.tolist()
        assert np.array_equal(_rows(eng, eng.x0), ref.x) and np.array_equal(_rows(eng, eng.r0), ref.r)
        eng.close()
    finally:
        for key, _, default in knobs:
            L.check(lib.mg_set_tuning(key.encode(), default))
