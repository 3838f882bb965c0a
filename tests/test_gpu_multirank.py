"""Several ranks sharing ONE GPU (gloo group, messages staged through the host): the multi-rank
schedule of the slab engine with the real HIP kernels, streams and the rank-0 tail graph.  Results
must equal the single-rank engine / the undecomposed oracle bit for bit.  Marked gpu."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, gny, gnx, n_para, h, b, cycles, fused, dist_min_rows, kw, q):
    from tests.mp_util import rank_watchdog
    rank_watchdog(int(os.environ.get("MG_TEST_RANK_WATCHDOG", "150")))
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from multigrid_amd.engine import SlabMultigrid
        # dead whole-domain levels every cycle for the larger grids: rank 0 runs them as a graph on a side stream
        eng = SlabMultigrid(gny, gnx, n_para, h, 0.0, world=world, rank=rank, fused=fused, eager_dead_levels=gny > 300,
                            dist_min_rows=dist_min_rows, recompute_min_nx=0, recompute_min_blocks=0, **(kw or {}))
        assert eng.comm.staged
        R = eng.plan.levels[0].rows
        a, e = eng.owned_rows()
        eng.set_b_rows(torch.from_numpy(b[rank * R : rank * R + (e - a)]).cuda())
        eng.presmooth()
        before = eng.comm.rounds
        errs = eng.cycles(cycles)
        assert eng.comm.rounds - before == cycles * eng.rounds_per_cycle(), (eng.comm.rounds - before, eng.rounds_per_cycle())
        if kw and kw.get("overlap_min_rows") == 1:  # the finest pass of every cycle at least was cut
            assert eng.split_passes >= cycles + 1 and eng._pending is None
        x = eng.local_rows(eng.x0).cpu().numpy()
        r = eng.local_rows(eng.r0).cpu().numpy()
        q.put((rank, errs, x, r))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,gny,gnx,n_para,fused", [(2, 257, 257, 3, True), (2, 129, 97, 1, True), (4, 1025, 513, 5, True),
                                                        (3, 769, 300 * 4 + 1, 2, True), (4, 1025, 513, 5, False)], ids=str)
@pytest.mark.parametrize("dist_min_rows", [8, 64])
def test_ranks_sharing_one_gpu_equal_oracle(world, gny, gnx, n_para, fused, dist_min_rows):
    _run(world, gny, gnx, n_para, fused, dist_min_rows)


# overlap_min_rows=1: passes that follow a halo exchange are cut into interior rows, launched beside the
# exchange (which runs on the communication stream), and the rows at the slab ends, launched after it
@pytest.mark.parametrize("world,gny,gnx,n_para,dist_min_rows,recompute",
                         [(2, 257, 257, 3, 8, True), (4, 1025, 513, 5, 8, True), (4, 1025, 513, 5, 64, False),
                          (3, 769, 1201, 2, 8, True), (2, 2049, 1025, 5, 8, True), (2, 2049, 1025, 5, 8, False)], ids=str)
def test_ranks_with_split_passes_equal_oracle(world, gny, gnx, n_para, dist_min_rows, recompute):
    _run(world, gny, gnx, n_para, True, dist_min_rows, dict(overlap_min_rows=1, recompute_presmoothing=recompute))


# red-black Gauss-Seidel on the fine levels (an option outside the reference): one launch per colour, one halo row
# exchanged after every colour
@pytest.mark.parametrize("world,gny,gnx,n_para,dist_min_rows", [(2, 257, 257, 3, 8), (4, 1025, 513, 5, 64), (3, 769, 1201, 2, 8)],
                         ids=str)
def test_ranks_with_redblack_fine_smoother_equal_oracle(world, gny, gnx, n_para, dist_min_rows):
    _run(world, gny, gnx, n_para, False, dist_min_rows, dict(fine_smoother="redblack"), dict(fine_smoother="redblack"))


def _run(world, gny, gnx, n_para, fused, dist_min_rows, kw=None, ref_kw=None):
    rng = np.random.default_rng(1)
    b = rng.standard_normal((gny, gnx))
    b[0] = b[-1] = 0
    b[:, 0] = b[:, -1] = 0
    h, cycles = 0.37, 3
    ref = oracle.GlobalSolver(gny, gnx, n_para, h, 0.0, **(ref_kw or {}))
    ref.b[:] = b
    ref.solve(ncycles=cycles)
    from tests.mp_util import run_ranks
    out = run_ranks(_worker, world, (gny, gnx, n_para, h, b, cycles, fused, dist_min_rows, kw))
    for k in range(world):
        assert out[k][0] == ref.err_hist.tolist()
    assert np.array_equal(np.concatenate([out[k][1] for k in range(world)]), ref.x)
    assert np.array_equal(np.concatenate([out[k][2] for k in range(world)]), ref.r)
