"""Host logic of the slab engine, checked without a GPU: the engine runs with the oracle-backed
ops (tests/oracle_ops.py) single-process and under torch.distributed/gloo with 2 and 4 ranks, and
must reproduce oracle.GlobalSolver (the reference schedule on the undecomposed grid) bit for bit:
Jacobi, residual and the transfer operators are order-free, so how the rows are cut cannot matter.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from multigrid_amd.engine import SlabMultigrid
from multigrid_amd.levels import exchanges_per_cycle, make_plan, reference_layout
from tests.oracle_ops import OracleOps
from tests.rhs import test_rhs as vortex_rhs


def _rhs(gny, gnx, seed=0):
    rng = np.random.default_rng(seed)
    b = rng.standard_normal((gny, gnx))
    b[0] = b[-1] = 0
    b[:, 0] = b[:, -1] = 0
    return b


def _engine_run(gny, gnx, n_para, h, b, world, rank, cycles, group=None, **kw):
    """b: global right-hand side; every rank takes its rows."""
    kw.setdefault("recompute_min_nx", 0)      # exercise the recompute scheme at these small sizes too
    kw.setdefault("recompute_min_blocks", 0)
    eng = SlabMultigrid(gny, gnx, n_para, h, 0.0, ops=OracleOps(), world=world, rank=rank, group=group, **kw)
    R = eng.plan.levels[0].rows
    a, e = eng.owned_rows()
    g0 = rank * R
    eng.set_b_rows(torch.from_numpy(b[g0 : g0 + (e - a)]))
    eng.presmooth()
    errs = eng.cycles(cycles)
    return eng, errs, eng.local_rows(eng.x0).numpy().copy(), eng.local_rows(eng.r0).numpy().copy()


CASES = [(65, 65, 2), (129, 129, 3), (65, 129, 3), (257, 33, 2), (33, 33, 1)]


@pytest.mark.parametrize("gny,gnx,n_para", CASES, ids=str)
@pytest.mark.parametrize("ordering,fused", [("lexicographic", True), ("lexicographic", False), ("redblack", True)])
def test_single_rank_equals_global_oracle(gny, gnx, n_para, ordering, fused):
    b = _rhs(gny, gnx)
    h = 0.37
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, ordering)
    ref.b[:] = b
    ref.solve(ncycles=4)
    eng, errs, x, r = _engine_run(gny, gnx, n_para, h, b, 1, 0, 4, coarse_ordering=ordering, fused=fused)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(x, ref.x)
    assert np.array_equal(r, ref.r)
    # whole-domain levels, dead ones included, match the oracle's
    eng.sync_whole_levels()
    for k in range(len(eng.plan.whole)):
        W = eng.plan.whole[k]
        assert np.array_equal(eng.xw[k].numpy()[:, : W.nx], ref.array("xw", k))
        assert np.array_equal(eng.rw[k].numpy()[:, : W.nx], ref.array("rw", k))


@pytest.mark.parametrize("fused", [True, False])
def test_damped_jacobi_equals_oracle_and_converges(fused):
    """jacobi_omega < 1 (an option of this build, not of the reference): same values as the oracle's
    damped schedule, and with the textbook ascent the cycle then converges on a right-hand side full
    of the checkerboard mode, where the undamped sweep stalls."""
    gny = gnx = 129
    b = _rhs(gny, gnx, 5)
    kw = dict(fix_coarse_ascent=True, jacobi_omega=0.8)
    ref = oracle.GlobalSolver(gny, gnx, 3, 0.5, 0.0, "lexicographic", **kw)
    ref.b[:] = b
    ref.solve(ncycles=6)
    eng, errs, x, r = _engine_run(gny, gnx, 3, 0.5, b, 1, 0, 6, fused=fused, **kw)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(x, ref.x) and np.array_equal(r, ref.r)
    und = oracle.GlobalSolver(gny, gnx, 3, 0.5, 0.0, "lexicographic", fix_coarse_ascent=True)
    und.b[:] = b
    und.solve(ncycles=6)
    assert ref.err_hist[-1] < 1e-4 * ref.err_hist[0] and und.err_hist[-1] > 0.05 * und.err_hist[0]


def test_fix_coarse_ascent_and_ascent_residuals():
    gny = gnx = 129
    b = _rhs(gny, gnx, 3)
    ref = oracle.GlobalSolver(gny, gnx, 2, 0.5, 0.0, "lexicographic", fix_coarse_ascent=True)
    ref.b[:] = b
    ref.solve(ncycles=3)
    eng, errs, x, _ = _engine_run(gny, gnx, 2, 0.5, b, 1, 0, 3, fix_coarse_ascent=True, keep_ascent_residuals=True)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(x, ref.x)
    L1 = eng.plan.levels[1]
    assert np.array_equal(eng.r[1].numpy()[L1.own0 : L1.own1, : L1.nx], ref.array("r", 1))
    # the textbook ascent converges faster than the reference's loop bounds allow
    ref2 = oracle.GlobalSolver(gny, gnx, 2, 0.5, 0.0, "lexicographic")
    ref2.b[:] = b
    ref2.solve(ncycles=3)
    assert ref.err_hist[-1] < ref2.err_hist[-1]


def test_solve_stop_rule_matches_oracle():
    n, n_para = 5, 2
    b0, h = vortex_rhs(n)
    N = 2 ** (n + 1) + 1
    ref = oracle.GlobalSolver(N, N, n_para, h, 0.025)
    ref.b[:] = b0
    e_ref = ref.solve()
    eng = SlabMultigrid(N, N, n_para, h, 0.025, ops=OracleOps(), world=1, rank=0)
    eng.set_b_rows(torch.from_numpy(b0))
    e = eng.solve()
    assert (e, eng.it) == (e_ref, ref.it)
    assert eng.err_history == ref.err_hist.tolist()
    # warm start: a second solve continues from the current x
    e2 = eng.solve()
    e2_ref = ref.solve()
    assert (e2, eng.it) == (e2_ref, ref.it)


def test_plan_and_layout():
    import json
    lay = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "layout.json")))
    for L in lay["levels"]:
        mine = reference_layout(L["n"], L["n_para"], 1.0)
        for k in ("x_sb", "b_sb", "r_sb", "x_wl", "b_wl", "h_sb", "h_wl"):
            assert mine[k] == L[k], (L["n"], L["n_para"], k)
    with pytest.raises(AssertionError):
        reference_layout(4, 4)
    p = make_plan(32769, 32769, 8, world=8, rank=3, h0=1.0)
    assert [L.rows for L in p.levels] == [4096 >> l for l in range(9)]
    assert p.levels[0].ld % 16 == 0 and p.levels[0].ld >= 32769
    assert (p.whole[0].ny, p.whole[0].nx) == (129, 129) and len(p.whole) == 7 and p.whole[-1].ny == 3
    assert p.halo == 8 and p.levels[0].t0 == 8 and make_plan(32769, 32769, 8, 8, 0).levels[0].t0 == 9
    assert make_plan(513, 513, 4).levels[0].t0 == 2 and make_plan(513, 513, 4).halo == 1
    with pytest.raises(ValueError):
        make_plan(129, 129, 5, world=4)  # 1 row per rank on the coarsest Jacobi level < halo
    with pytest.raises(ValueError):
        make_plan(65, 65, 2, world=3)
    with pytest.raises(ValueError):
        make_plan(65, 65, 6)
    with pytest.raises(ValueError):
        make_plan(67, 65, 2)
    assert exchanges_per_cycle(4, fused=False) == 8 + 33 + 1 and exchanges_per_cycle(4) == 8


# ------------------------------------------------------------------ multi-rank, gloo


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gny, gnx, n_para, h, b, cycles, q, fused=True, dist_min_rows=None, **kw):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng, errs, x, r = _engine_run(gny, gnx, n_para, h, b, world, rank, cycles, fused=fused,
                                      dist_min_rows=dist_min_rows, **kw)
        # communication rounds of one more cycle (its last pass also stores r: no extra round)
        before = eng.comm.rounds
        eng.cycles(1)
        assert eng.comm.rounds - before == eng.rounds_per_cycle(), (eng.n_dist, eng.comm.rounds - before)
        assert eng.rounds_per_cycle() <= exchanges_per_cycle(eng.n_dist, fused)
        if kw.get("overlap_min_rows") == 1:  # the finest pass of every cycle at least is cut (and presmooth's)
            assert eng.split_passes >= cycles + 1 and eng._pending is None
            q.put(("splits", rank, eng.split_passes))
            # solve() with a cap of as many cycles (epsilon = 0 is never reached) is the same sequence
            eng2 = SlabMultigrid(gny, gnx, n_para, h, 0.0, ops=OracleOps(), world=world, rank=rank, fused=fused,
                                 dist_min_rows=dist_min_rows, max_cycles=cycles, recompute_min_nx=0,
                                 recompute_min_blocks=0, **kw)
            a, e = eng2.owned_rows()
            eng2.set_b_rows(torch.from_numpy(b[rank * eng2.plan.levels[0].rows :][: e - a]))
            assert eng2.solve() == errs[-1] and eng2.it == cycles and eng2._pending is None
            assert np.array_equal(eng2.local_rows(eng2.x0).numpy(), x)
            assert np.array_equal(eng2.local_rows(eng2.r0).numpy(), r)
        q.put((rank, errs, x, r))
        dist.barrier()
    finally:
        dist.destroy_process_group()


# dist_min_rows: Jacobi levels with fewer rows per rank are replicated on every rank instead of being
# cut into slabs (8 = the halo depth: as many distributed levels as possible; 256 = the default, which
# at these sizes leaves only the finest level distributed)
@pytest.mark.parametrize("dist_min_rows", [8, 16, None])
@pytest.mark.parametrize("world,gny,gnx,n_para,fused", [(2, 65, 65, 2, True), (2, 65, 33, 1, True), (2, 129, 65, 3, True), (2, 129, 65, 3, False),
                                                        (4, 129, 129, 3, True), (4, 257, 33, 2, True),
                                                        (4, 257, 33, 2, False),
                                                        (8, 1025, 65, 4, True)], ids=str)
def test_gloo_ranks_equal_global_oracle(world, gny, gnx, n_para, fused, dist_min_rows):
    _gloo_run_equals_oracle(world, gny, gnx, n_para, fused, dist_min_rows)


# overlap_min_rows=1: every pass that follows a halo exchange is cut into interior rows (enqueued before
# the exchange is waited for) and the rows at both slab ends, wherever the slab is tall enough for the
# cut; right-hand-side halos are exchanged by the pass that reads them first.  Same values, same rounds.
@pytest.mark.parametrize("world,gny,gnx,n_para,dist_min_rows,recompute",
                         [(2, 129, 65, 3, None, True), (2, 257, 129, 3, 8, True), (4, 257, 33, 2, 8, True),
                          (8, 1025, 65, 4, 8, True), (2, 513, 33, 4, 8, True), (2, 513, 33, 4, 8, False),
                          (4, 513, 65, 3, 8, False)], ids=str)
def test_gloo_ranks_with_split_passes_equal_global_oracle(world, gny, gnx, n_para, dist_min_rows, recompute):
    splits = _gloo_run_equals_oracle(world, gny, gnx, n_para, True, dist_min_rows, overlap_min_rows=1,
                                     recompute_presmoothing=recompute)
    if (gny - 1) // world >= 256:  # tall slabs: coarser levels are cut too (descent and ascent passes)
        assert min(splits.values()) > 2 * 5


# dist_levels: the caller's own choice of how many levels are cut into slabs (bench.py --dist-levels), and the cost
# model fed with measured prices (t_round, gather_bw; bench.py times both at start-up): whatever is chosen, same values
@pytest.mark.parametrize("kw", [dict(dist_levels=1), dict(dist_levels=2), dict(dist_levels=9), dict(t_round=1e-9, gather_bw=1e6),
                                dict(t_round=1.0, gather_bw=1e15)], ids=str)
def test_gloo_ranks_with_a_given_distribution_depth_equal_global_oracle(kw):
    _gloo_run_equals_oracle(2, 257, 65, 4, True, None, **kw)


def test_distribution_depth_follows_the_prices_it_is_given(monkeypatch):
    from multigrid_amd.levels import choose_n_dist
    # free rounds and a slow all-gather: cut every level that can be cut; dear rounds and a free all-gather: only the finest
    assert choose_n_dist(8 * 32768 + 1, 32769, 11, 8, 12, t_round=1e-9, gather_bw=1e6) >= 8
    assert choose_n_dist(8 * 32768 + 1, 32769, 11, 8, 12, t_round=1.0, gather_bw=1e15) == 1
    monkeypatch.setenv("MG_T_ROUND_US", "1000000")
    monkeypatch.setenv("MG_GATHER_GBPS", "1000000")
    assert choose_n_dist(8 * 32768 + 1, 32769, 11, 8, 12) == 1
    monkeypatch.delenv("MG_T_ROUND_US")
    monkeypatch.delenv("MG_GATHER_GBPS")
    assert choose_n_dist(8 * 32768 + 1, 32769, 11, 8, 12) == 6  # the defaults' choice at the bench's weak-scaling size


def _gloo_run_equals_oracle(world, gny, gnx, n_para, fused, dist_min_rows, **kw):
    b = _rhs(gny, gnx, 7)
    h = 0.37
    cycles = 3
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0)
    ref.b[:] = b
    ref.solve(ncycles=cycles)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, world, port, gny, gnx, n_para, h, b, cycles, q, fused, dist_min_rows),
                         kwargs=kw) for rk in range(world)]
    for p in procs:
        p.start()
    out, splits = {}, {}
    while len(out) < world:
        msg = q.get(timeout=240)
        if msg[0] == "splits":
            splits[msg[1]] = msg[2]
            continue
        rank, errs, x, r = msg
        out[rank] = (errs, x, r)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = np.concatenate([out[k][1] for k in range(world)], axis=0)
    r = np.concatenate([out[k][2] for k in range(world)], axis=0)
    for k in range(world):
        assert out[k][0] == ref.err_hist.tolist()  # every rank sees the global max-norm
    assert np.array_equal(x, ref.x)
    assert np.array_equal(r, ref.r)
    return splits


def test_batch_of_problems_equals_independent_solves():
    """batch=B: B independent problems in one engine == B separate runs of the undecomposed oracle."""
    B, N, n_para, h = 3, 65, 2, 0.37
    rng = np.random.default_rng(4)
    bs = rng.standard_normal((B, N, N))
    bs[:, 0] = bs[:, -1] = 0
    bs[:, :, 0] = bs[:, :, -1] = 0
    refs = []
    for k in range(B):
        ref = oracle.GlobalSolver(N, N, n_para, h, 0.0)
        ref.b[:] = bs[k]
        ref.solve(ncycles=3)
        refs.append(ref)
    eng = SlabMultigrid(N, N, n_para, h, 0.0, ops=OracleOps(), world=1, rank=0, batch=B)
    eng.set_b_rows(torch.from_numpy(bs))
    eng.presmooth()
    errs = eng.cycles(3)
    assert errs == [max(r.err_hist[c] for r in refs) for c in range(3)]
    x, r = eng.local_rows(eng.x0).numpy(), eng.local_rows(eng.r0).numpy()
    for k in range(B):
        assert np.array_equal(x[k], refs[k].x)
        assert np.array_equal(r[k], refs[k].r)


def test_choose_n_para_keeps_the_whole_domain_level_in_lds():
    from multigrid_amd.levels import choose_n_para, make_plan, whole_level_fits_one_wave
    for world, want in ((1, 8), (2, 9), (4, 10), (8, 11)):
        gny, gnx = world * 2 ** 15 + 1, 2 ** 15 + 1
        n_para = choose_n_para(gny, gnx, world, at_least=8)
        assert n_para == want
        W0 = make_plan(gny, gnx, n_para, world, 0).whole[0]
        assert whole_level_fits_one_wave(W0.ny, W0.nx)
        if n_para > 8:
            W0 = make_plan(gny, gnx, n_para - 1, world, 0).whole[0]
            assert not whole_level_fits_one_wave(W0.ny, W0.nx)
    assert choose_n_para(9, 9, 1, at_least=1) == 1
    # bench.py's default: from n - 4 levels up (the reference's own proportion, a 33 x 33 whole-domain level)
    for world, want in ((1, 10), (2, 10), (4, 10), (8, 11)):
        assert choose_n_para(world * 2 ** 15 + 1, 2 ** 15 + 1, world, at_least=10) == want
    assert choose_n_para(2 ** 15 + 1, 2 ** 15 + 1, 8, at_least=10) == 10  # strong scaling: 4096 rows per rank


def test_choose_n_dist_weighs_rounds_against_the_all_gather():
    """The benchmark grids: levels are distributed until the all-gather of the next level's right-hand
    side (and the redundant work below it) costs less than two more exchange rounds."""
    from multigrid_amd.levels import choose_n_dist, choose_n_para
    for world, want in ((2, 5), (4, 6), (8, 6)):
        gny, gnx = world * 2 ** 15 + 1, 2 ** 15 + 1
        n_para = choose_n_para(gny, gnx, world, at_least=8)
        k = choose_n_dist(gny, gnx, n_para, world, 12)
        assert k == want
        assert ((gny - 1) // world >> (k - 1)) >= 12
        # the gathered level is small: at most 32 MB
        assert (((gny - 1) >> k) + 1) * (((gnx - 1) >> k) + 1) * 8 <= 32e6
    assert choose_n_dist(257, 257, 3, 2, 12) == 1          # tiny grid: replicate everything below the finest
    assert choose_n_dist(65, 65, 2, 4, 8) == 1             # slabs of 16 rows: the second level would hold fewer than the halo
    # free exchanges and a slow gather: distribute every level whose slabs are tall enough
    assert choose_n_dist(2 ** 15 + 1, 2 ** 15 + 1, 8, 2, 12, t_round=0.0, gather_bw=1e6) == 8


def test_cut_passes_partition_rows_and_stay_clear_of_the_halo():
    """The arithmetic of SlabMultigrid._pass at the benchmark's sizes (which no CPU run reaches): the three
    launches of a cut pass partition the updated rows, their coarse rows partition the coarse level's, every
    coarse row's fine centre lies in its own launch, and the interior launch reads owned rows only -- of the
    fine arrays (D rows up and down) and of the coarse correction (the rows the prolongation touches)."""
    # weak scaling (32768 rows per rank) and, last, strong scaling (32768 rows in all)
    for world, n_para, n_dist, rows in ((2, 10, 5, None), (4, 10, 6, None), (8, 11, 6, None), (8, 8, 8, None), (8, 10, 4, 2 ** 15)):
        gny, gnx = (rows or world * 2 ** 15) + 1, 2 ** 15 + 1
        for rank in range(world):
            plan = make_plan(gny, gnx, n_para, world, rank, 1.0, halo=12, n_dist=n_dist)
            H = plan.halo
            for l in range(n_dist):
                L, Lc = plan.levels[l], plan.levels[l + 1]
                for D in (3, 5, 6, 8, 9, 10):
                    if L.rows < 4 * (D + 4):
                        continue
                    a, e = SlabMultigrid._cut(L, D)
                    parts = [(L.t0, a), (a, e), (e, L.t1)]
                    assert L.t0 < a < e < L.t1 and (a - H) % 2 == 0 and (e - H) % 2 == 0
                    # interior: fine rows read are owned and are not the rows next to the halo
                    assert a - D > L.own0 and e + D - 1 < H + L.rows - 1
                    # ... and so are the coarse rows under them: fine row t reads coarse rows (t+H)>>1 and +1
                    assert ((a - D) + H) >> 1 >= H and (((e + D - 1) + H) >> 1) + 1 <= H + Lc.rows - 1
                    T = [SlabMultigrid._coarse_rows(Lc, H, j0, j1) for j0, j1 in parts]
                    assert T[0][0] == Lc.t0 and T[2][1] == Lc.t1 and T[0][1] == T[1][0] and T[1][1] == T[2][0]
                    for (j0, j1), (T0, T1) in zip(parts, T):
                        assert T0 < T1 and j0 <= 2 * T0 - H and 2 * (T1 - 1) - H < j1


def test_choose_n_dist_only_proposes_plans_make_plan_accepts():
    """The cost model ranks only slab depths whose plan exists: rows per rank divisible by 2**k and at least
    `halo` rows on the coarsest distributed level (sizes with odd multiples of a power of two included)."""
    from multigrid_amd.levels import choose_n_dist, make_plan
    checked = 0
    for world in (2, 3, 4, 8):
        for cells in (400, 96, 1536, 200 * world, 48 * world, 4096, 2 ** 15, 3 * 2 ** 10):
            for n_para in (1, 2, 3, 4, 6):
                gny = cells + 1
                try:
                    make_plan(gny, 257, n_para, world, 0, halo=8, n_dist=1)
                except ValueError:
                    continue
                for halo in (2, 8, 12):
                    if (cells // world) < halo:
                        continue
                    k = choose_n_dist(gny, 257, n_para, world, halo)
                    make_plan(gny, 257, n_para, world, 0, halo=halo, n_dist=k)  # must not raise
                    checked += 1
    assert checked > 50


# ---------------------------------------------------------------------------------------------------------
# placement selection (SlabMultigrid._place_finest): the choice logic, driven with scripted timings
class _FakeArr:
    def __init__(self, k):
        self.k = k

    def data_ptr(self):
        return self.k << 33


class _PlacementRig:
    """Stands in for the engine, its ops and torch.cuda's events inside _place_finest: every candidate allocation
    is a numbered object, a timed launch costs what `cost(b, x_in, x_out)` says, and no launch may alias two roles."""

    def __init__(self, cost, max_allocs=64):
        import types
        self.cost, self.max_allocs = cost, max_allocs
        self.count, self.last = 0, None
        lvl = types.SimpleNamespace(ny=32769, ld=32784, nx=32769, t0=1, t1=32768, fix_lo=0, fix_hi=32768, oj=-1, h=1.0)
        lvc = types.SimpleNamespace(ny=16385, ld=16400, nx=16385, t0=1, t1=16384)
        self.lv = [lvl, lvc]
        self.plan = types.SimpleNamespace(levels=[lvl, lvc])
        self.N2, self.jacobi_omega = 6, 1.0
        self.ops = self
        self.device = None
        self.stream = self
        self.placement = None
        for name in ("PLACE_MAX_CANDIDATES", "PLACE_SPREAD_CANDIDATES"):
            setattr(self, name, getattr(SlabMultigrid, name))
        self.PLACE_BUDGET_SECONDS = 1e9
        self._spread_plan = SlabMultigrid._spread_plan

    # ops
    def zeros(self, rows, ld, batch=None):
        if rows == 32769:
            if self.count >= self.max_allocs:
                raise torch.OutOfMemoryError("scripted")
            self.count += 1
            return _FakeArr(self.count - 1)
        return _FakeArr(-1)

    def norm_slots(self):
        return _FakeArr(-2)

    def begin(self):
        pass

    def zero_(self, a):
        pass

    def synchronize(self):
        pass

    def smooth_fused(self, b, x_in, x_out, r, crs, *a, **k):
        assert b is not x_in and b is not x_out and x_in is not x_out, "two roles of one launch in one array"
        self.last = (b.k, x_in.k, x_out.k)

    def _place_level1(self, *a, **k):
        return None


def _run_placement(monkeypatch, cost, max_allocs=64, budget=1e9, free_bytes=280 << 30):
    rig = _PlacementRig(cost, max_allocs)
    rig.PLACE_BUDGET_SECONDS = budget

    class Ev:
        def __init__(self, enable_timing=False):
            pass

        def record(self, stream):
            pass

        def elapsed_time(self, other):
            return 2.0 * rig.cost(*rig.last)

    monkeypatch.setattr(torch.cuda, "Event", Ev)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (free_bytes, 288 << 30))
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
    out = SlabMultigrid._place_finest(rig)
    return rig, out


@pytest.mark.parametrize("budget", [0.0, 1e9])
@pytest.mark.parametrize("n_slow,max_allocs", [(3, 4), (3, 64), (11, 64), (5, 6), (0, 64), (40, 64)])
def test_placement_selection_never_puts_two_roles_in_one_array(monkeypatch, n_slow, max_allocs, budget):
    """Scripted timings: the first n_slow candidates are slow as outputs, later ones fast; a launch that reads its
    right-hand side and x from one stretch of the sequence is a little slower.  Whatever ends the candidate loop
    (memory, the time budget, enough fast ones), the four chosen arrays are four different allocations and no timed
    launch ever had one array in two roles (with one fast candidate among slow ones, the spread step used to try the
    array chosen as b as an x as well)."""
    def cost(b, xi, xo):
        return (6.1 if xo < n_slow else 5.4) + (0.1 if (b < n_slow) == (xi < n_slow) else 0.0)

    rig, out = _run_placement(monkeypatch, cost, max_allocs, budget)
    if out is None:
        return
    fin, cb, xa1 = out
    assert len({a.k for a in fin}) == 4
    assert rig.placement["chosen"] == [a.k for a in fin]


def test_spread_plan_skips_the_arrays_already_in_a_role():
    plan = SlabMultigrid._spread_plan
    # b's candidate is also the second x of the four chosen: it is not tried as an x
    assert plan([3, 0, 1, 2], [0, 1, 2], [4, 5, 6, 7]) == (3, 0, [6, 7])
    # the fastest output is itself of the slow kind and first in line for b: nothing to spread
    assert plan([0, 3, 1, 2], [0, 1, 2], [4, 5, 6, 7]) == (0, 0, [])
    assert plan([5, 4, 6, 7], [0, 1], []) == (5, 0, [4])
