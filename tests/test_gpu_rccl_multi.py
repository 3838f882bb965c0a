"""The slab engine over real RCCL: one process per GPU, `nccl` backend, 2 ranks and as many as there are devices
(at most 8): the Python engine launching pass by pass (its exchanges raw RCCL groups of the library), the same with a
rank's whole cycle captured into one hipGraph, and the C handle mg_dist_*.  Runs wherever at least two devices are visible and is skipped on a one-GPU lease -- there the same
schedule is covered by the gloo tests (ranks sharing the GPU, tests/test_gpu_multirank.py) and the one-rank RCCL
calls (tests/test_gpu_rccl.py).  Every rank keeps 4096 rows of a 4097-column grid, so the multi-rank grid is
(world * 4096 + 1) x 4097; every pass that follows an exchange is cut around it (overlap_min_rows = 1).  Results must
equal the undecomposed oracle bit for bit, and the number of communication rounds the engine's own count.
Replaces the per-sweep exchange of Buffers.exchange_buffers (core/parallel_multigrid.py:83-97), the four broadcasts of
gather_blocks (:117-138) and the vote (:342-345).  Marked gpu."""
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

ROWS, GNX, N_PARA, CYCLES = 4096, 4097, 7, 2
# MG_TEST_RCCL_REHEARSE=1: run the same test with gloo ranks sharing device 0 (checks the test itself on a one-GPU box)
REHEARSE = os.environ.get("MG_TEST_RCCL_REHEARSE") == "1"


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


def _worker(rank, world, port, use_graph, q, mode="engine"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if REHEARSE:  # the test's own logic on one GPU: the ranks share device 0, messages staged through the host
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from multigrid_amd.engine import SlabMultigrid
        gny = world * ROWS + 1
        h = 20.0 / (GNX - 1)
        rng = np.random.default_rng(100 + rank)  # every rank makes its own rows; the parent rebuilds the same grid
        rows = rng.standard_normal((ROWS + (1 if rank == world - 1 else 0), GNX))
        rows[:, 0] = rows[:, -1] = 0
        if rank == 0:
            rows[0] = 0
        if rank == world - 1:
            rows[-1] = 0
        if mode == "handle":  # the C handle: mg_dist_* on the group's communicator, a rank's cycle one graph launch
            from multigrid_amd.solver import NativeDistSolver
            s = NativeDistSolver.from_torch_group(gny, GNX, N_PARA, h, use_graph=use_graph)
            assert s.info()["owned_rows"] == rows.shape[0] and s.info()["first_row"] == rank * ROWS
            s.upload("b", rows)
            s.presmooth()
            errs = s.cycles(CYCLES)
            before = s.info()["rounds"]
            errs += s.cycles(1)
            rounds = s.info()["rounds"] - before
            q.put((rank, errs, s.download("x"), s.download("r"), rounds, rounds, s.info()["n_dist"], 1))
            dist.barrier()
            s.close()
            return
        eng = SlabMultigrid(gny, GNX, N_PARA, h, 0.0, world=world, rank=rank, eager_dead_levels=True, overlap_min_rows=1,
                            use_graph=use_graph, graph_comm=mode == "graph_comm")
        assert eng.comm.staged == REHEARSE and dist.get_world_size() == world
        assert eng.graph_comm == (mode == "graph_comm" and use_graph and not REHEARSE)
        a, e = eng.owned_rows()
        assert e - a == rows.shape[0]
        eng.set_b_rows(torch.from_numpy(rows).cuda())
        eng.presmooth()
        errs = eng.cycles(CYCLES)
        before = eng.comm.rounds
        errs += eng.cycles(1)
        rounds = eng.comm.rounds - before
        x = eng.local_rows(eng.x0).cpu().numpy()
        r = eng.local_rows(eng.r0).cpu().numpy()
        per_cycle = eng.rounds_per_cycle()
        if eng.graph_comm:  # the rounds are inside the graph, which is not cut: nothing to count here
            rounds, per_cycle, cut = 0, 0, 1
        else:
            cut = eng.split_passes
        q.put((rank, errs, x, r, rounds, per_cycle, eng.n_dist, cut))
        dist.barrier()
        eng.close()
    finally:
        dist.destroy_process_group()


def _worker_mode(rank, world, port, use_graph, mode, q):
    _worker(rank, world, port, use_graph, q, mode)


def _worlds():
    if REHEARSE:
        return [2, 3]
    n = min(8, _device_count())
    return sorted({w for w in (2, n) if 2 <= w <= n})


@pytest.mark.skipif(_device_count() < 2 and not REHEARSE, reason="needs at least two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("use_graph,mode", [(True, "engine"), (False, "engine"), (True, "graph_comm"), (True, "handle"), (False, "handle")],
                         ids=["graph", "eager", "whole-cycle-graph", "c-handle-graph", "c-handle-eager"])
def test_slab_engine_over_rccl_equals_oracle(use_graph, mode):
    from tests.mp_util import run_ranks
    if REHEARSE and mode == "handle":
        pytest.skip("the C handle talks to RCCL only")
    for world in _worlds():
        out = run_ranks(_worker_mode, world, (use_graph, mode), timeout=280)
        gny = world * ROWS + 1
        b = np.zeros((gny, GNX))
        for rank in range(world):
            rng = np.random.default_rng(100 + rank)
            n_own = ROWS + (1 if rank == world - 1 else 0)
            rows = rng.standard_normal((n_own, GNX))
            rows[:, 0] = rows[:, -1] = 0
            if rank == 0:
                rows[0] = 0
            if rank == world - 1:
                rows[-1] = 0
            b[rank * ROWS : rank * ROWS + n_own] = rows
        ref = oracle.GlobalSolver(gny, GNX, N_PARA, 20.0 / (GNX - 1), 0.0)
        ref.b[:] = b
        ref.solve(ncycles=CYCLES + 1)
        for rank in range(world):
            errs, x, r, rounds, per_cycle, n_dist, cut = out[rank]
            assert errs == ref.err_hist.tolist(), (world, rank)
            assert rounds == per_cycle and n_dist >= 1, (world, rank, rounds, per_cycle)
            assert cut > 0  # passes were in fact cut around their exchanges
        assert np.array_equal(np.concatenate([out[k][1] for k in range(world)]), ref.x)
        assert np.array_equal(np.concatenate([out[k][2] for k in range(world)]), ref.r)
