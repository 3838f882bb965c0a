"""The RCCL branches of SlabComm on a one-rank `nccl` group: RCCL refuses two ranks on one
device, so the calls the multi-GPU engine makes -- a batch of row sends/receives, the all-gather of
whole rows into a padded array, the integer MAX-reduction of the norm slots -- are issued here with
the rank as its own neighbour, on views shaped like the engine's: once through torch.distributed ("direct")
and once as the raw RCCL calls libmg_hip.so makes on the group's communicator ("native", the engine's default).  Checks that the backend accepts
the tensors as they are (views, dtypes) and moves the right bytes, and that the engine's hipGraph
capture and replay work in a process that holds an RCCL communicator.  Marked gpu."""
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(port, q):
    import time

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from multigrid_amd.comm import SlabComm, torch_nccl_comm
        from multigrid_amd.ops import HipOps
        ops = HipOps(torch.device("cuda", 0))
        H, rows, ld = 12, 64, 144

        def exercise(comm):
            """The engine's three communication patterns on `comm`, the rank being its own neighbour."""
            ops.begin()
            with ops.ctx():
                x = torch.arange((rows + 2 * H) * ld, dtype=torch.float64, device="cuda").view(rows + 2 * H, ld)
                want = x.clone()
                # the engine's exchange: own rows next to the halo go out, the halo rows come in
                # (here the lower neighbour's send is this rank's own, so halo_lo <- top own rows and so on)
                sends = [(x[H:2 * H].view(-1), 0), (x[rows:rows + H].view(-1), 0)]
                recvs = [(x[rows + H:].view(-1), 0), (x[:H].view(-1), 0)]
                want[rows + H:] = want[H:2 * H]
                want[:H] = want[rows:rows + H]
            batch = comm.prepare(sends, recvs)
            comm.run(batch)
            comm.run(batch)  # a prepared batch is issued every cycle
            # the same batch on the communication stream, beside a kernel of the main sequence (split passes)
            with ops.ctx():
                scratch = torch.ones(1 << 20, dtype=torch.float64, device="cuda")
            ready = ops.mark()
            ops.zero_(scratch)
            with ops.comm_side(ready):
                comm.run(batch)
            ops.comm_join()
            with ops.ctx():
                inp = x[H:H + rows]
                out = torch.zeros(rows + 5, ld, dtype=torch.float64, device="cuda")
            comm.allgather_rows(out[2:2 + rows], inp)
            with ops.ctx():
                slots = ops.norm_slots()
                slots[3] = 0x7FF8000000000000  # a NaN's bit pattern stays the maximum
                slots[5] = 12345
                before = slots.clone()
            comm.allreduce_max(slots)
            ops.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                comm.run(batch)
            ops.stream.synchronize()
            per_round = (time.perf_counter() - t0) / 50
            ops.end()
            ok = (bool(torch.equal(x, want)) and bool(scratch.eq(0).all()), bool(torch.equal(out[2:2 + rows], inp)),
                  bool(out[:2].eq(0).all() and out[2 + rows:].eq(0).all()), bool(torch.equal(slots, before)))
            return ok, per_round, batch, scratch, out, inp

        comm = SlabComm(ops, None, 1, 0)  # torch.distributed's calls (one rank: nothing attached)
        assert not comm.staged and comm.native is None
        ok_direct, per_round, batch, scratch, out, inp = exercise(comm)
        # the same through the library's raw RCCL calls on the group's communicator
        ncomm = SlabComm(ops, None, 1, 0)
        ncomm._attach(torch_nccl_comm(None, ops.device))
        assert ncomm.native is not None and ncomm.mode == "native"
        ok_native, per_round_native, nbatch, _, _, _ = exercise(ncomm)
        # a message table the library must refuse: a received block that overlaps a sent one; a peer outside the communicator
        from multigrid_amd._lib import MgError
        with ops.ctx():
            y = torch.zeros(4 * ld, dtype=torch.float64, device="cuda")
        refused = 0
        for sends_, recvs_ in (([(y[: 2 * ld], 0)], [(y[ld : 3 * ld], 0)]), ([(y[:ld], 1)], [(y[ld : 2 * ld], 0)])):
            try:
                ncomm.prepare(sends_, recvs_)
            except MgError:
                refused += 1
        ok_native = ok_native + (refused == 2,)
        # capturing and replaying the cycle graphs while the RCCL communicator and its helper threads are alive
        import numpy as np
        from multigrid_amd.engine import SlabMultigrid
        rng = np.random.default_rng(0)
        b = rng.standard_normal((257, 257))
        b[0] = b[-1] = 0
        b[:, 0] = b[:, -1] = 0
        runs = []
        for graph in (True, False):
            eng = SlabMultigrid(257, 257, 3, 0.3, 0.0, world=1, rank=0, use_graph=graph, eager_dead_levels=True)
            eng.set_b_rows(torch.from_numpy(b).cuda())
            eng.presmooth()
            errs = eng.cycles(3)
            comm.run(batch)  # communication between graph launches
            errs += eng.cycles(2)
            runs.append((errs, eng.local_rows(eng.x0).cpu().numpy()))
            eng.close()
        graphs_ok = runs[0][0] == runs[1][0] and np.array_equal(runs[0][1], runs[1][1])
        # a capture that begins right after a collective on the same stream and stays open while torch's watchdog
        # thread polls that collective's end event (every 100 ms): the multi-GPU engine captures the replicated
        # middle of the cycle right after the all-gather.  HipOps records on a stream of its own for this reason;
        # on the collective's stream the poll fails with hipErrorCapturedEvent and ends the process.
        for _ in range(3):
            ops.begin()
            for _ in range(20):
                ops.zero_(scratch)
            comm.allgather_rows(out[2:2 + rows], inp)
            ops.graph_begin()
            ops.zero_(scratch)
            time.sleep(0.35)
            ops.zero_(scratch)
            g = ops.graph_end()
            ops.graph_launch(g)
            ops.stream.synchronize()
            ops.graph_destroy(g)
            ops.end()
        # The whole multi-rank engine over these RCCL calls, the rank being its own neighbours (values are meaningless
        # then; that every call is accepted, the counts, and that nothing ends the process are not): cut passes on
        # the communication stream, both slab ends in one launch, the all-gather, the replicated middle and the dead
        # levels captured into graphs right after RCCL traffic, solve()'s order of norm reads, close().
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
        from loopback_comm import LoopbackComm
        engine_runs = []
        for world, rank in ((2, 0), (4, 2), (8, 7)):
            gny, gnx = world * 1024 + 1, 1025
            ops2 = HipOps(torch.device("cuda", 0))
            lc = LoopbackComm(ops2, world, rank, rccl=True, native=world != 4)  # the library's raw RCCL calls; torch's once
            assert lc.mode == ("native" if world != 4 else "direct")
            eng = SlabMultigrid(gny, gnx, 6, 0.01, 0.0, ops=ops2, world=world, rank=rank, comm=lc, eager_dead_levels=True,
                                overlap_min_rows=1, dist_min_rows=64, max_cycles=3)
            a, e = eng.owned_rows()
            eng.set_b_rows(torch.from_numpy(rng.standard_normal((e - a, gnx))).cuda())
            eng.presmooth()
            r0 = lc.rounds
            errs = eng.cycles(3)
            per_cycle = (lc.rounds - r0) / 3
            eng.solve()
            # (solve() may stop before its third cycle: with the rank's own rows as halo the divergence latch can trip)
            engine_runs.append((len(errs), bool(np.isfinite(errs).all()), per_cycle, eng.rounds_per_cycle(), eng.split_passes > 0,
                                1 <= eng.it <= 3, eng.n_dist >= 2))
            eng.close()
        # the same rank with its whole cycle, exchanges included, captured into one hipGraph per cycle (graph_comm):
        # the raw RCCL calls are recorded like kernels; values equal the launch-by-launch run bit for bit
        pair = []
        for gc in (False, True):
            ops3 = HipOps(torch.device("cuda", 0))
            lc = LoopbackComm(ops3, 8, 3, rccl=True, native=True)
            eng = SlabMultigrid(8 * 1024 + 1, 1025, 6, 0.01, 0.0, ops=ops3, world=8, rank=3, comm=lc, eager_dead_levels=True,
                                overlap_min_rows=1, dist_min_rows=64, graph_comm=gc)
            assert eng.graph_comm == gc and (eng.overlap_min_rows is None) == gc
            a, e = eng.owned_rows()
            eng.set_b_rows(torch.from_numpy(np.random.default_rng(7).standard_normal((e - a, 1025))).cuda())
            eng.presmooth()
            errs = eng.cycles(2)
            errs += eng.cycles(3)
            pair.append((errs, eng.local_rows(eng.x0).cpu().numpy(), eng.local_rows(eng.r0).cpu().numpy()))
            eng.close()
        graph_comm_ok = pair[0][0] == pair[1][0] and np.array_equal(pair[0][1], pair[1][1]) and np.array_equal(pair[0][2], pair[1][2])
        # the C handle of the several-rank solver (mg_dist_*), rehearsed the same way (MG_DIST_LOOPBACK): the same calls on
        # the same rows in the same order as the engine with its whole cycle in one graph, so the same (meaningless) values
        from multigrid_amd.comm import loaded_rccl_path
        from multigrid_amd.solver import NativeDistSolver
        import ctypes as C
        from multigrid_amd import _lib
        handle_ok = True
        for world, rank, gny_, gnx_, npara_ in ((8, 3, 8 * 1024 + 1, 1025, 6), (2, 0, 2 * 2048 + 1, 513, 5), (4, 3, 4 * 512 + 1, 2049, 4)):
            ops4 = HipOps(torch.device("cuda", 0))
            lc = LoopbackComm(ops4, world, rank, rccl=True, native=True)
            eng = SlabMultigrid(gny_, gnx_, npara_, 0.01, 0.0, ops=ops4, world=world, rank=rank, comm=lc, eager_dead_levels=True,
                                dist_min_rows=64, graph_comm=True)
            a, e = eng.owned_rows()
            rows_b = np.random.default_rng(11 + world).standard_normal((e - a, gnx_))
            eng.set_b_rows(torch.from_numpy(rows_b).cuda())
            eng.presmooth()
            errs_e = eng.cycles(3)
            x_e, r_e = eng.local_rows(eng.x0).cpu().numpy(), eng.local_rows(eng.r0).cpu().numpy()
            n_dist = eng.n_dist
            eng.close()
            c = C.c_void_p()
            path = loaded_rccl_path()
            _lib.check(_lib.lib().mg_comm_attach(C.c_void_p(torch_nccl_comm(None, ops.device)), path.encode() if path else None, C.byref(c)))
            for graph in (True, False):
                hs = NativeDistSolver(gny_, gnx_, npara_, 0.01, c, rank, world, dist_levels=n_dist, loopback=True, use_graph=graph)
                assert hs.info()["n_dist"] == n_dist and hs.info()["owned_rows"] == e - a
                hs.upload("b", rows_b)
                hs.presmooth()
                errs_h = hs.cycles(3)
                same = errs_h == errs_e and np.array_equal(hs.download("x"), x_e) and np.array_equal(hs.download("r"), r_e)
                handle_ok = handle_ok and same
                print(f"C handle, rank {rank} of {world}, graph {graph}: equals the engine: {same}", flush=True)
                hs.close()
            _lib.lib().mg_comm_destroy(c)
        graph_comm_ok = graph_comm_ok and handle_ok
        print("loopback engine runs:", engine_runs, "whole-cycle graph with exchanges equals eager:", graph_comm_ok, flush=True)
        graphs_ok = graphs_ok and graph_comm_ok
        graphs_ok = graphs_ok and all(r[0] == 3 and r[1] and r[2] == r[3] and r[4] and r[5] and r[6] for r in engine_runs)
        ncomm.close()
        q.put((ok_direct + (graphs_ok,), ok_native, per_round, per_round_native))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_slabcomm_direct_branch_on_one_rank_rccl_group():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    try:
        ok, ok_native, per_round, per_round_native = q.get(timeout=150)
    finally:
        p.join(timeout=30)
        if p.is_alive():
            p.kill()
    print(f"one exchange round on a one-rank RCCL group: {per_round * 1e6:.0f} us through torch.distributed, "
          f"{per_round_native * 1e6:.0f} us as raw RCCL calls of the library")
    assert ok == (True, True, True, True, True)
    assert ok_native == (True, True, True, True, True)
    assert p.exitcode == 0
