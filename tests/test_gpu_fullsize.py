"""Parity at the benchmark's own sizes (marked gpu).

The CPU oracle takes seconds per V-cycle at 16385^2 and the level arrays take 46 GB at 32769^2, so the
largest grid is covered through properties that do not need the oracle:
  * two independent code paths of this package -- fused passes inside one hipGraph vs. one kernel per
    operation, launched eagerly -- agree bit for bit;
  * the stored residual equals b - A x recomputed by the single-operation kernel, and the reported
    norm equals its max-norm;
  * scaling the right-hand side by a power of two scales the solution by exactly that factor
    (every operation of the cycle is linear and a power-of-two factor commutes with rounding).
The engine is also compared with the oracle itself, bit for bit, at 4097^2, at 16385^2 and (where the host has the
memory for the oracle's arrays) at 32769^2 with the benchmark's own level layout."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def _engine(n, n_para, **kw):
    import torch
    import bench
    from multigrid_amd.engine import SlabMultigrid
    N = 2 ** (n + 1) + 1
    eng = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0, **kw)
    bench.fill_rhs(eng, torch, n, 0)
    return eng


def test_fused_graph_equals_separate_kernels_at_32769():
    import torch
    n, n_para = 14, 8
    # tune_placement: set-up times candidate allocations as the output of the finest pass and keeps the fastest
    # as the x arrays; they must come out of it zeroed and laid out like any others
    a = _engine(n, n_para, eager_dead_levels=True, tune_placement=True)
    assert a.placement is not None and a.placement["candidates"] >= 4 and a.placement_seconds < 10
    a.presmooth()
    ea = a.cycles(2)
    xa = a.x0.clone()
    ra = a.r0.clone()
    a.close()
    del a
    torch.cuda.empty_cache()
    b = _engine(n, n_para, fused=False, use_graph=False)
    b.presmooth()
    eb = b.cycles(2)
    assert ea == eb
    assert torch.equal(xa, b.x0)
    assert torch.equal(ra, b.r0)
    # the stored residual is b - A x, and the reported norm its max-norm
    L = b.plan.levels[0]
    chk = torch.zeros_like(ra)
    b.ops.begin()
    b.ops.residual(chk, b.x0, b.b0, L.nx, L.t0, L.t1, L.h)
    b.ops.end()
    torch.cuda.synchronize()
    assert torch.equal(chk, ra)
    assert float(ra.abs().max()) == ea[-1]
    b.close()


def test_default_engine_selects_its_placement_at_32769():
    """A caller who passes no switch gets the placement selection wherever the finest arrays are large enough for it to
    matter (>= 1 GiB each): the finest pass on the chosen arrays is of the fast kind, and set-up stays short."""
    e = _engine(14, 10, eager_dead_levels=True)
    assert e.placement is not None and e.placement["candidates"] >= 4 and e.placement_seconds < 10
    assert e.placement["ms_chosen_combination"] <= 1.06 * min(e.placement["ms_as_output"])  # the two kinds are 10 % and more apart
    e.close()
    small = _engine(10, 6)
    assert small.placement is None
    small.close()


def test_power_of_two_scaling_is_exact_at_32769():
    import torch
    n, n_para = 14, 8
    e = _engine(n, n_para)
    e.presmooth()
    e1 = e.cycles(2)
    x1 = e.x0.clone()
    # same engine, right-hand side times 4, x back to zero
    e.b0.mul_(4.0)
    e.set_b_rows(e.local_rows(e.b0).clone())
    e.set_x_rows(torch.zeros_like(e.local_rows(e.x0)))
    e.presmooth()
    e4 = e.cycles(2)
    assert e4 == [4.0 * v for v in e1]
    assert torch.equal(e.x0, 4.0 * x1)
    e.close()


def test_engine_equals_oracle_at_16385():
    import torch
    n, n_para = 13, 7
    e = _engine(n, n_para, eager_dead_levels=True)
    N = e.plan.gnx
    ref = oracle.GlobalSolver(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0)
    ref.b[:] = e.local_rows(e.b0).cpu().numpy()
    ref.solve(ncycles=2)
    e.presmooth()
    errs = e.cycles(2)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(e.local_rows(e.x0).cpu().numpy(), ref.x)
    assert np.array_equal(e.local_rows(e.r0).cpu().numpy(), ref.r)
    e.close()


def _equal_rows(dev_rows, host, chunk=2048):
    """dev_rows (a device view) == host (numpy), compared in row chunks so that no second full-size host copy exists."""
    for a in range(0, host.shape[0], chunk):
        if not np.array_equal(dev_rows[a : a + chunk].cpu().numpy(), host[a : a + chunk]):
            return False
    return True


@pytest.mark.parametrize("graph,recompute", [(True, None), (True, True), (True, False), (False, True)],
                         ids=["graph-default", "graph-recompute", "graph-stored", "eager-recompute"])
def test_engine_equals_oracle_at_4097(graph, recompute):
    """The 4096^2 configuration (n = 11, seven Jacobi levels, 33 x 33 first whole-domain level) against the oracle,
    bit for bit: the captured cycle with the default choice of levels with recomputed pre-smoothing, with the
    recompute scheme forced on every level, with it off, and launched eagerly."""
    n, n_para = 11, 7
    kw = {}
    if recompute is not None:
        kw = dict(recompute_presmoothing=recompute, recompute_min_nx=0, recompute_min_blocks=0)
    e = _engine(n, n_para, eager_dead_levels=True, use_graph=graph, **kw)
    N = e.plan.gnx
    assert N == 4097 and e.plan.whole[0].nx == 33
    ref = oracle.GlobalSolver(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0)
    ref.b[:] = e.local_rows(e.b0).cpu().numpy()
    ref.solve(ncycles=2)
    e.presmooth()
    errs = e.cycles(2)
    assert errs == ref.err_hist.tolist()
    assert np.array_equal(e.local_rows(e.x0).cpu().numpy(), ref.x)
    assert np.array_equal(e.local_rows(e.r0).cpu().numpy(), ref.r)
    e.ops.synchronize()
    for k, W in enumerate(e.plan.whole):
        assert np.array_equal(e.xw[k].cpu().numpy()[:, : W.nx], ref.array("xw", k))
    e.close()


def test_engine_equals_oracle_at_32769():
    """The benchmark's own workload (32769 x 32769, ten Jacobi levels, dead whole-domain levels computed every
    cycle, one hipGraph per cycle) against the oracle, bit for bit: two cycles' max|r|, all of x and all of r.
    The oracle holds 43 GB of level arrays on the host and needs a few seconds per cycle."""
    import psutil
    need = 70 << 30
    if psutil.virtual_memory().available < need:
        pytest.skip(f"the CPU oracle needs about 45 GB of host memory at this size; {psutil.virtual_memory().available >> 30} GiB available")
    n, n_para = 14, 10
    e = _engine(n, n_para, eager_dead_levels=True)
    N = e.plan.gnx
    assert N == 32769 and e.plan.whole[0].nx == 33
    ref = oracle.GlobalSolver(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0)
    brows = e.local_rows(e.b0)
    for a in range(0, N, 2048):
        ref.b[a : a + 2048] = brows[a : a + 2048].cpu().numpy()
    ref.solve(ncycles=2)
    e.presmooth()
    errs = e.cycles(2)
    assert errs == ref.err_hist.tolist()
    assert _equal_rows(e.local_rows(e.x0), ref.x)
    assert _equal_rows(e.local_rows(e.r0), ref.r)
    e.ops.synchronize()
    for k, W in enumerate(e.plan.whole):
        assert np.array_equal(e.xw[k].cpu().numpy()[:, : W.nx], ref.array("xw", k))
    e.close()


def test_batch_of_1024_problems_equals_oracle_on_sampled_problems():
    """The batched configuration at its full size (1024 independent 257 x 257 problems, four Jacobi levels):
    every problem gets its own right-hand side (the vortex pattern scaled and shifted per problem); a sample
    of problems is compared with the oracle bit for bit, the norm with the maximum over the sample's."""
    import torch
    from multigrid_amd.engine import SlabMultigrid
    from tests.rhs import test_rhs
    n, n_para, B = 7, 4, 1024
    N = 2 ** (n + 1) + 1
    b0, h = test_rhs(n)
    b0[0] = b0[-1] = 0
    b0[:, 0] = b0[:, -1] = 0
    eng = SlabMultigrid(N, N, n_para, h, 0.0, batch=B, eager_dead_levels=True)
    scale = torch.linspace(0.5, 1.5, B, dtype=torch.float64, device="cuda")
    base = torch.from_numpy(b0).cuda()
    rows = base.unsqueeze(0) * scale.view(B, 1, 1)
    shift = torch.arange(B, device="cuda") % 64
    for k in range(64):  # problem p sees the pattern rolled by p % 64 columns
        sel = shift == k
        rows[sel] = torch.roll(rows[sel], k, dims=2)
    rows[:, :, 0] = 0
    rows[:, :, -1] = 0
    eng.set_b_rows(rows)
    eng.presmooth()
    errs = eng.cycles(3)
    x = eng.local_rows(eng.x0)
    r = eng.local_rows(eng.r0)
    worst = 0.0
    for p in (0, 1, 63, 64, 517, 1023):
        ref = oracle.GlobalSolver(N, N, n_para, h, 0.0)
        ref.b[:] = rows[p].cpu().numpy()
        ref.solve(ncycles=3)
        assert np.array_equal(x[p].cpu().numpy(), ref.x), p
        assert np.array_equal(r[p].cpu().numpy(), ref.r), p
        assert ref.err_hist[-1] <= errs[-1]
        worst = max(worst, ref.err_hist[-1])
    assert float(r.abs().max()) == errs[-1] >= worst
    per = eng.problem_errors()
    assert per.shape == (B,) and float(per.max()) == errs[-1]
    assert float(per[517]) == float(r[517].abs().max())
    eng.close()


def _two_rank_worker(rank, world, port, n, n_para, cycles, q):  # started by tests.mp_util.run_ranks
    import torch
    import torch.distributed as dist
    import bench
    from multigrid_amd.engine import SlabMultigrid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gnx = 2 ** (n + 1) + 1
        eng = SlabMultigrid(world * 2 ** (n + 1) + 1, gnx, n_para, 20.0 / 2 ** (n + 1), 0.0, world=world, rank=rank,
                            eager_dead_levels=True)
        bench.fill_rhs(eng, torch, n, rank)
        eng.presmooth()
        errs = eng.cycles(cycles)
        a, e = eng.owned_rows()
        x = eng.local_rows(eng.x0)
        picks = [0, 1, 7, 8, 9, 13, 14, 15, 16, 17, 100, (e - a) // 2, e - a - 18, e - a - 17, e - a - 16, e - a - 15, e - a - 9,
                 e - a - 8, e - a - 2, e - a - 1]
        q.put((rank, errs, eng.n_dist, eng.split_passes, picks, x[picks].cpu().numpy(),
               eng.local_rows(eng.r0)[picks].cpu().numpy()))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_at_the_benchmark_slab_size_equal_one_rank():
    """The multi-GPU bench's own slab size (32768 x 32769 per rank, five distributed levels, the four largest
    with their passes cut around the halo exchange) on two ranks that share the GPU (gloo, messages staged
    through the host): per-cycle norms equal those of one rank on the same 65537 x 32769 grid, and so do rows of
    x and r at the slab ends, next to them and inside."""
    import torch
    import bench
    from multigrid_amd.engine import SlabMultigrid
    from tests.mp_util import run_ranks
    n, n_para, world, cycles = 14, 10, 2, 2
    out = run_ranks(_two_rank_worker, world, (n, n_para, cycles), timeout=200)
    one = SlabMultigrid(world * 2 ** (n + 1) + 1, 2 ** (n + 1) + 1, n_para, 20.0 / 2 ** (n + 1), 0.0, world=1, rank=0,
                        eager_dead_levels=True)
    bench.fill_rhs(one, torch, n, 0)
    one.presmooth()
    errs = one.cycles(cycles)
    x, r = one.local_rows(one.x0), one.local_rows(one.r0)
    R = 2 ** (n + 1)
    for rank in range(world):
        e_k, n_dist, cut, picks, xs, rs = out[rank]
        assert e_k == errs
        # cut: the opening pass, and per cycle the finest pass and the descent and ascent passes of levels 1-3
        # (not the first descent pass of level 1: presmooth() ends with every halo exchanged)
        assert n_dist == 5 and cut == 7 * cycles
        rows = [rank * R + t for t in picks]
        assert np.array_equal(xs, x[rows].cpu().numpy())
        assert np.array_equal(rs, r[rows].cpu().numpy())
    one.close()


# ---------------------------------------------------------------------------------------------------------
# The grids of the multi-GPU configurations cut into as many slabs as they name, every rank a process of its own
# sharing the one GPU (gloo, messages staged through the host), values checked.

def _scratch_dir(need_bytes):
    """A directory for arrays the spawned ranks read: in memory where that has the room, else under the temporary directory."""
    import shutil
    import tempfile
    base = None
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) and shutil.disk_usage("/dev/shm").free > need_bytes + (1 << 30):
        base = "/dev/shm"
    elif shutil.disk_usage(tempfile.gettempdir()).free < need_bytes + (1 << 30):
        pytest.skip(f"no room for {need_bytes >> 20} MiB of arrays shared with the spawned ranks")
    return tempfile.mkdtemp(prefix="mg_slabs_", dir=base)


def _slab_worker(rank, world, port, n, n_para, cycles, ddir, kw, q):  # started by tests.mp_util.run_ranks
    """Rank `rank` of `world` on the (2^(n+1)+1)^2 grid cut into `world` slabs.  ddir holds b.npy (the global right-hand
    side) and, when the caller has them, x.npy / r.npy (what every rank's rows must equal bit for bit)."""
    from tests.mp_util import rank_watchdog
    rank_watchdog(int(os.environ.get("MG_TEST_RANK_WATCHDOG", "170")))
    import torch
    import torch.distributed as dist
    from multigrid_amd.engine import SlabMultigrid
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N = 2 ** (n + 1) + 1
        eng = SlabMultigrid(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0, world=world, rank=rank, eager_dead_levels=True, **kw)
        R = eng.plan.levels[0].rows
        a, e = eng.owned_rows()
        g0 = rank * R
        b = np.load(os.path.join(ddir, "b.npy"), mmap_mode="r")
        eng.set_b_rows(torch.from_numpy(np.ascontiguousarray(b[g0 : g0 + (e - a)])).cuda())
        eng.presmooth()
        before = eng.comm.rounds
        errs = eng.cycles(cycles)
        rounds_ok = eng.comm.rounds - before == cycles * eng.rounds_per_cycle()
        x, r = eng.local_rows(eng.x0), eng.local_rows(eng.r0)
        same = None
        if os.path.exists(os.path.join(ddir, "x.npy")):
            xr = np.load(os.path.join(ddir, "x.npy"), mmap_mode="r")
            rr = np.load(os.path.join(ddir, "r.npy"), mmap_mode="r")
            same = (_equal_rows(x, xr[g0 : g0 + (e - a)]), _equal_rows(r, rr[g0 : g0 + (e - a)]))
        # rows at the slab ends, next to them, around the cuts of the passes and inside
        m = e - a
        picks = sorted({t for t in (0, 1, 2, 7, 8, 9, 11, 12, 13, 15, 16, 17, 18, 19, 23, 24, 25, 100, m // 2, m - 26, m - 25, m - 20, m - 19,
                                    m - 18, m - 17, m - 16, m - 14, m - 13, m - 12, m - 9, m - 8, m - 3, m - 2, m - 1) if 0 <= t < m})
        q.put((rank, errs, eng.n_dist, eng.split_passes, rounds_ok, same, [g0 + t for t in picks], x[picks].cpu().numpy(),
               r[picks].cpu().numpy()))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_16385_grid_in_eight_slabs_equals_oracle():
    """16385 x 16385 cut into EIGHT slabs of 2048 rows (nine Jacobi levels, as many of them distributed as the cost
    model chooses, a 33 x 33 first whole-domain level): every rank's per-cycle norms, all of x and all of r equal
    the undecomposed oracle bit for bit, with the expected number of communication rounds."""
    import shutil
    import torch
    from tests.mp_util import run_ranks
    n, n_para, world, cycles = 13, 9, 8, 2
    one = _engine(n, n_para, eager_dead_levels=True)  # for the right-hand side (built on the device)
    N = one.plan.gnx
    b = one.local_rows(one.b0).cpu().numpy()
    one.close()
    del one
    torch.cuda.empty_cache()
    ref = oracle.GlobalSolver(N, N, n_para, 20.0 / 2 ** (n + 1), 0.0)
    ref.b[:] = b
    ref.solve(ncycles=cycles)
    ddir = _scratch_dir(3 * b.nbytes)
    try:
        np.save(os.path.join(ddir, "b.npy"), b)
        np.save(os.path.join(ddir, "x.npy"), ref.x)
        np.save(os.path.join(ddir, "r.npy"), ref.r)
        out = run_ranks(_slab_worker, world, (n, n_para, cycles, ddir, {}), timeout=220)
    finally:
        shutil.rmtree(ddir, ignore_errors=True)
    for rank in range(world):
        errs, n_dist, cut, rounds_ok, same, rows, xs, rs = out[rank]
        assert errs == ref.err_hist.tolist(), rank
        assert same == (True, True), rank
        assert rounds_ok and 1 <= n_dist <= n_para, rank
        assert np.array_equal(xs, ref.x[rows]) and np.array_equal(rs, ref.r[rows])


@pytest.mark.parametrize("world", [4, 8])
def test_32769_grid_in_slabs_equals_one_rank(world):
    """32769 x 32769 (ten Jacobi levels) cut into 4 and into 8 slabs, the partitions of the strong-scaling runs: every
    rank's per-cycle norms equal those of the engine on one rank (itself equal to the oracle at this size,
    test_engine_equals_oracle_at_32769), and so do rows of x and r at the slab ends, next to them, around the cuts of
    the passes and inside; the largest levels' passes are cut around their halo exchange."""
    import shutil
    import torch
    from tests.mp_util import run_ranks
    n, n_para, cycles = 14, 10, 2
    one = _engine(n, n_para, eager_dead_levels=True)
    ddir = _scratch_dir(one.plan.gnx ** 2 * 8)
    try:
        brows = one.local_rows(one.b0)
        N = one.plan.gnx
        bm = np.lib.format.open_memmap(os.path.join(ddir, "b.npy"), mode="w+", dtype=np.float64, shape=(N, N))
        for a in range(0, N, 2048):
            bm[a : a + 2048] = brows[a : a + 2048].cpu().numpy()
        bm.flush()
        del bm
        one.presmooth()
        errs1 = one.cycles(cycles)
        x1, r1 = one.local_rows(one.x0), one.local_rows(one.r0)
        R = (N - 1) // world
        m = R
        want = {}
        for rank in range(world):
            mm = m + (1 if rank == world - 1 else 0)
            picks = sorted({t for t in (0, 1, 2, 7, 8, 9, 11, 12, 13, 15, 16, 17, 18, 19, 23, 24, 25, 100, mm // 2, mm - 26, mm - 25, mm - 20,
                                        mm - 19, mm - 18, mm - 17, mm - 16, mm - 14, mm - 13, mm - 12, mm - 9, mm - 8, mm - 3, mm - 2, mm - 1)
                            if 0 <= t < mm})
            rows = [rank * R + t for t in picks]
            want[rank] = (rows, x1[rows].cpu().numpy(), r1[rows].cpu().numpy())
        one.close()
        del one, x1, r1, brows
        torch.cuda.empty_cache()
        out = run_ranks(_slab_worker, world, (n, n_para, cycles, ddir, {}), timeout=240)
    finally:
        shutil.rmtree(ddir, ignore_errors=True)
    for rank in range(world):
        errs, n_dist, cut, rounds_ok, same, rows, xs, rs = out[rank]
        assert errs == errs1, rank
        assert rounds_ok and n_dist >= 2 and cut >= 1 + cycles, (rank, n_dist, cut)
        assert rows == want[rank][0]
        assert np.array_equal(xs, want[rank][1]) and np.array_equal(rs, want[rank][2]), rank
