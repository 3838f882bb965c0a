#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One step = one V-cycle of the reference schedule (3 pre / 6 post Jacobi sweeps per Jacobi level,
3 symmetric SOR sweeps on the first whole-domain level, residual + max-norm on the finest level,
convergence value read back by the host) on a right-hand side that is already resident in HBM.

Workload: the configuration BASELINE.json's metric is quoted on -- a 32769 x 32769 float64 grid
(n = 14, ~1.07e9 unknowns, 46 GB of level arrays) per GPU.  With N > 1 ranks (one process per GPU,
launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py
--gpus N`, before anything touches a GPU) every rank keeps that slab and the global grid grows to
(N * 32768 + 1) x 32769 rows x columns: weak scaling.  The halo rows of the distributed levels travel by RCCL
send/recv; how many of the Jacobi levels are cut into slabs is decided by a cost model of one cycle
(levels.choose_n_dist: 5, 6, 6 at N = 2, 4, 8; `config.distributed_levels` reports it); the levels below are
all-gathered once per cycle and computed redundantly on every rank.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      the dominant kernel (the finest level's fused pass: prolongation, 6 Jacobi sweeps, residual,
                max-norm and restriction in one launch).  `achieved` = the bytes that launch itself must move
                (28 B per point: it reads x, b and the coarse x, writes x and the coarse b, each once) / its
                average duration, timed here with HIP events on the stream the kernel runs on; `frac` = that
                over the 8 TB/s HBM peak.  `algorithmic_equivalent` prices the same launch by the bytes the
                operations it performs would move one by one (196 B per point): a throughput equivalent, not
                a bandwidth.  `traffic` = HBM bytes per launch from rocprofv3 FETCH_SIZE/WRITE_SIZE counters
                (profiles/traffic.json, measured on tools/pmc_probe.py; printed only when the kernel sources
                are the ones it was measured on).  The plain Jacobi sweep (24 B per point) is beside it.
  cycle         the same two prices for the whole V-cycle
  host_issue_ms_per_cycle, per_rank   host time to enqueue one cycle (clock stopped before the device is waited for), and every
                rank's own record gathered to rank 0: its ms per step, host issue time, finest pass and Jacobi sweep ms,
                communication rounds per cycle, passes cut around their exchange, communication mode
  config.comm   N > 1: how the rows travel ("native" = raw RCCL groups issued by libmg_hip.so on the engine's streams) and
                the exchange round and all-gather timed at start-up, from which the number of distributed levels was chosen
  pcie          host <-> device copy of one finest-level array, measured after the timed region (never part of `value`)
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm) timed on this host's cores on a
                bounded sample: the undecomposed schedule with OpenMP row-parallel sweeps on all cores, and
                the reference's own decomposition -- four 2 x 2 blocks, one thread per block, an exchange per
                sweep -- at n = 8 .. 11 (`four_ranks_2x2`)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# RCCL and CUDA-tensor sharing between processes need dmabuf IPC handles on hosts whose driver supports no other kind
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--exp", dest="n", type=int, default=14, help="grid is (2**(n+1)+1) columns wide; 14 -> 32769")
    ap.add_argument("--jacobi-levels", dest="n_para", type=int, default=0,
                    help="number of Jacobi levels; 0 = n - 4, the proportion of the reference's own run (n = 8, n_para = 4: "
                         "a 33 x 33 first whole-domain level), or more where that level, which grows with the rank "
                         "count, would leave the single-wave SSOR kernel otherwise (10, 10, 10, 11 on 1, 2, 4, 8 ranks at n = 14)")
    ap.add_argument("--ordering", default="lexicographic", choices=["lexicographic", "redblack"])
    ap.add_argument("--roofline-iters", type=int, default=20)
    ap.add_argument("--cpu-exp", dest="cpu_n", type=int, default=12, help="grid exponent of the CPU baseline sample")
    ap.add_argument("--cpu-cycles", type=int, default=0, help="0 = choose for roughly 15 s")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lazy-dead-levels", action="store_true",
                    help="do not run the whole-domain levels whose result the reference schedule never uses")
    ap.add_argument("--no-fused", action="store_true", help="one kernel per operation instead of fused passes")
    ap.add_argument("--no-recompute", action="store_true",
                    help="coarse levels store x after the pre-smoothing instead of recomputing it in the post-smoothing pass")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the (2**(n+1)+1)^2 grid is cut into N slabs instead of growing by N times "
                         "the rows (the default, weak scaling)")
    ap.add_argument("--no-tune-placement", action="store_true",
                    help="allocate the level arrays where the allocator puts them instead of timing a few "
                         "placements of the finest level's arrays at set-up and keeping the fastest")
    ap.add_argument("--no-untuned-leg", action="store_true",
                    help="skip the few cycles on arrays without placement selection that are reported beside the result")
    ap.add_argument("--batch", type=int, default=0,
                    help="B > 0: B independent (2**(n+1)+1)^2 problems advance together on one GPU "
                         "(e.g. --batch 1024 --exp 7 --jacobi-levels 4); not the headline workload")
    ap.add_argument("--jacobi-omega", type=float, default=1.0,
                    help="damping of the Jacobi sweeps (1 = the reference's sweep; an option of this build)")
    ap.add_argument("--dist-levels", type=int, default=0,
                    help="N > 1: how many of the Jacobi levels (the finest ones) are cut into slabs; 0 = the cost model's choice "
                         "(levels.choose_n_dist) from the exchange round and the all-gather timed at start-up "
                         "(or MG_T_ROUND_US / MG_GATHER_GBPS from the environment)")
    ap.add_argument("--overlap-min-rows", type=int, default=4096,
                    help="N > 1: passes of levels with at least this many rows per rank are cut into interior rows and slab ends "
                         "around their halo exchange")
    ap.add_argument("--graph-comm", action="store_true",
                    help="N > 1: the whole V-cycle of a rank, halo exchanges and all-gather included (raw RCCL calls recorded by "
                         "hipGraph capture), is one graph launch per cycle instead of one launch per pass and exchange")
    ap.add_argument("--no-comm-calibration", action="store_true",
                    help="N > 1: do not time an exchange round and an all-gather at start-up; the cost model's defaults decide")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host <-> device copy of one finest-level array")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages messages through the host: lets several ranks share one GPU (rehearsal only)")
    return ap.parse_args()


def fill_rhs(eng, torch, n, rank):
    """The eight-vortex right-hand side of the reference's test driver (test/test.py:40-58), built on
    the device slab by slab.  Rows beyond the first 20-unit band repeat the pattern so that every
    rank's slab carries the same kind of data."""
    P = eng.plan
    gnx = P.gnx
    h = 20.0 / 2 ** (n + 1)
    dev = eng.b0.device
    X = torch.linspace(-10, 10, gnx, dtype=torch.float64, device=dev)
    xr = [4, -2, 6, 3, 3, 5, -5, -7]
    yr = [-2, 6, 6, -5, 5, -5, -2, -2]
    a, e = eng.owned_rows()
    band = 2 ** (n + 1)
    chunk = 256
    for t in range(a, e, chunk):
        t2 = min(t + chunk, e)
        g = torch.arange(t, t2, device=dev, dtype=torch.float64) + (rank * P.levels[0].rows - P.halo)
        Y = (-10.0 + h * torch.remainder(g, band)).unsqueeze(1)
        acc = torch.zeros((t2 - t, gnx), dtype=torch.float64, device=dev)
        sign = 1.0
        for xv, yv in zip(xr, yr):
            acc += sign * 50.0 * torch.exp(-7.0 * ((X.unsqueeze(0) - xv) ** 2 + (Y - yv) ** 2))
            sign = -sign
        eng.b0[..., t:t2, :gnx] = acc
    # Dirichlet boundary cells carry no equation
    eng.b0[..., 0] = 0
    eng.b0[..., gnx - 1] = 0
    L0 = P.levels[0]
    if rank == 0:
        eng.b0[..., L0.own0, :] = 0
    if rank == eng.world - 1:
        eng.b0[..., L0.own1 - 1, :] = 0
    eng.set_b_rows(eng.local_rows(eng.b0))  # marks the halo of b as stale
    return h


def cpu_baseline(args):
    import numpy as np
    import oracle
    from tests.rhs import test_rhs
    n = args.cpu_n
    N = 2 ** (n + 1) + 1
    # keep n - n_para (hence the size of the whole-domain level) as in the GPU workload
    n_para = max(1, min(n - 1, args.n_para - (args.n - n)))
    b0, h = test_rhs(n)
    s = oracle.GlobalSolver(N, N, n_para, h, 0.0, "lexicographic")
    s.b[:] = b0
    s.solve(ncycles=1)  # first touch of the level arrays
    t0 = time.perf_counter()
    s.solve(ncycles=2)
    t1 = (time.perf_counter() - t0) / 2
    cycles = args.cpu_cycles or max(2, min(400, int(20.0 / max(t1, 1e-3))))
    t0 = time.perf_counter()
    s.solve(ncycles=cycles)
    dt = time.perf_counter() - t0
    unknowns = (N - 2) * (N - 2)
    out = {
        "value": unknowns * cycles / dt, "unit": "unknowns*V-cycles/s", "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"{cycles} V-cycles on a {N}x{N} grid (n={n}, n_para={n_para}), C restatement of the reference "
                  f"algorithm with OpenMP row-parallel sweeps, {dt:.1f} s",
        "ms_per_cycle": 1e3 * dt / cycles,
    }
    # The reference's own decomposition: four blocks (2 x 2) with a duplicated centre line, one worker per block
    # as its `mpiexec -n 4` (core/Makefile:7, test/Makefile:9), a halo exchange after every sweep.  n_para = n - 4
    # keeps the 33 x 33 first whole-domain level of its own n = 8, n_para = 4 run.
    legs = []
    for nb in (8, 9, 10, 11):
        bb, hb = test_rhs(nb)
        blk = oracle.BlockSolver(nb, nb - 4, hb, 0.0, rank_parallel=True)
        for k in range(4):
            oracle.split(bb, blk.b[k], k)
        blk.set_max_cycles(2)
        blk.solve()  # first touch
        cyc = {8: 60, 9: 40, 10: 16, 11: 6}[nb]
        blk.set_max_cycles(cyc)
        t0 = time.perf_counter()
        blk.solve()
        db = time.perf_counter() - t0
        unk = (2 ** (nb + 1) - 1) ** 2
        legs.append({"n": nb, "n_para": nb - 4, "grid": 2 ** (nb + 1) + 1, "cycles": blk.it, "ms_per_cycle": 1e3 * db / blk.it,
                     "value": unk * blk.it / db})
    out["four_ranks_2x2"] = {"unit": "unknowns*V-cycles/s", "cores": 4, "kind": "port",
                             "sample": "oracle.BlockSolver (the reference's four 2x2 blocks, one thread per block, exchange "
                                       "per sweep, test.py right-hand side) at n = 8..11, n_para = n - 4",
                             "runs": legs}
    return out


def calibrate_comm(torch, dist, world, rank, gnx, halo):
    """Time what the distribution cost model prices, on this machine, before the engine is built: one halo exchange
    round of the engine's shape (`halo` rows of a gnx-wide array to and from each neighbour) and one all-gather of
    an 8 MiB slab per rank, each issued alone on the stream (GPU time per call by HIP events, host time per call by
    the clock).  Every rank gets the same numbers: the slowest rank's round, the slowest all-gather."""
    from multigrid_amd.comm import SlabComm
    from multigrid_amd.levels import pitch
    from multigrid_amd.ops import HipOps
    ops = HipOps()
    comm = SlabComm(ops, None, world, rank)
    ld = pitch(gnx)
    R = 64
    arr = ops.zeros(R + 2 * halo, ld)
    sends, recvs = [], []
    if rank > 0:
        sends.append((arr[halo : 2 * halo].view(-1), rank - 1))
        recvs.append((arr[:halo].view(-1), rank - 1))
    if rank < world - 1:
        sends.append((arr[R : R + halo].view(-1), rank + 1))
        recvs.append((arr[R + halo :].view(-1), rank + 1))
    batch = comm.prepare(sends, recvs)
    grows = max(1, (8 << 20) // (ld * 8))
    gin, gout = ops.zeros(grows, ld), ops.zeros(world * grows, ld)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def timed(fn, reps):
        ops.begin()
        fn()
        fn()
        ops.stream.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        ev[0].record(ops.stream)
        for _ in range(reps):
            fn()
        host = (time.perf_counter() - t0) / reps
        ev[1].record(ops.stream)
        ops.stream.synchronize()
        ops.end()
        return host, ev[0].elapsed_time(ev[1]) * 1e-3 / reps

    round_host, round_gpu = timed(lambda: comm.run(batch), 20)
    gather_host, gather_gpu = timed(lambda: comm.allgather_rows(gout, gin), 10)
    t = torch.tensor([round_host, round_gpu, gather_host, gather_gpu], dtype=torch.float64,
                     device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    round_host, round_gpu, gather_host, gather_gpu = (float(v) for v in t)
    gbytes = grows * ld * 8 * (world - 1)  # what arrives at one rank
    mode = comm.mode
    comm.close()
    return {"comm_mode": mode, "round_host_us": 1e6 * round_host, "round_gpu_us": 1e6 * round_gpu,
            "round_bytes_per_neighbour": halo * ld * 8, "allgather_host_us": 1e6 * gather_host, "allgather_gpu_us": 1e6 * gather_gpu,
            "allgather_bytes_received": gbytes, "allgather_gbps": gbytes / max(gather_gpu, 1e-9) / 1e9,
            # what choose_n_dist is given: a round costs the cycle whichever is longer, issuing it or running it
            "t_round_us": 1e6 * max(round_host, round_gpu), "gather_gbps": gbytes / max(gather_gpu, gather_host, 1e-9) / 1e9}


def pcie_copy(torch, eng):
    """Host <-> device copy of ONE finest-level array, measured: what a caller whose arrays live in host memory pays
    per array on top of the cycles (the reference's level-0 arrays are the caller's own numpy arrays,
    core/parallel_multigrid.py:196-198).  The owned rows of b go device -> host through a pinned staging buffer of
    at most 1 GiB, chunk by chunk; the same number of bytes then goes host -> device into a scratch array."""
    rows = eng.local_rows(eng.b0)
    ny, nx = rows.shape[-2], rows.shape[-1]
    chunk = max(1, min(ny, (1 << 30) // (nx * 8)))
    stage = torch.empty((chunk, nx), dtype=torch.float64).pin_memory()
    scratch = torch.empty((chunk, nx), dtype=torch.float64, device=rows.device)
    total = ny * nx * 8
    secs = {}
    for name in ("d2h", "h2d"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a in range(0, ny, chunk):
            m = min(chunk, ny - a)
            if name == "d2h":
                stage[:m].copy_(rows[a : a + m], non_blocking=True)
            else:
                scratch[:m].copy_(stage[:m], non_blocking=True)
            torch.cuda.synchronize()
        secs[name] = time.perf_counter() - t0
    del stage, scratch
    return {"what": "the owned rows of one finest-level array (b) device -> host and the same bytes host -> device, through a "
                    "pinned staging buffer, chunks of at most 1 GiB",
            "bytes": total, "d2h_ms": 1e3 * secs["d2h"], "h2d_ms": 1e3 * secs["h2d"],
            "d2h_gbps": total / secs["d2h"] / 1e9, "h2d_gbps": total / secs["h2d"] / 1e9}


def main():
    args = parse()
    # Libraries write to the process's standard output too (RCCL prints a version banner there when it makes its
    # communicator): everything but the one JSON line goes to standard error from here on.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched by torch.distributed.run (one rank per GPU)")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        # a rank that is still here after ten minutes writes every thread's stack to standard error and ends: a hang among
        # the ranks then shows where each of them waits instead of running into the launcher's time limit without a word
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ.get("MG_BENCH_WATCHDOG_S", "600")), exit=True, file=sys.stderr)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")

    from multigrid_amd.engine import SlabMultigrid
    from multigrid_amd.levels import algorithmic_bytes_per_cycle, own_bytes_per_cycle

    from multigrid_amd.levels import choose_n_para
    n = args.n
    gnx = 2 ** (n + 1) + 1
    gny = (1 if args.strong else world) * 2 ** (n + 1) + 1
    # as many Jacobi levels as leave the reference's own coarse size (test/test.py: n = 8, n_para = 4, a 33 x 33
    # first whole-domain level): the chain of small passes is a little longer, the serial SSOR much shorter
    n_para = args.n_para or choose_n_para(gny, gnx, world, at_least=max(1, n - 4))
    if not n_para:
        sys.exit(f"no admissible number of Jacobi levels for a {gny}x{gnx} grid on {world} ranks")
    args.n_para = n_para
    h = 20.0 / 2 ** (n + 1)

    # N > 1: what an exchange round and the all-gather cost here decides how many levels are cut into slabs
    calib = None
    if world > 1 and not args.no_comm_calibration and not args.dist_levels:
        try:
            calib = calibrate_comm(torch, dist, world, rank, gnx, 12 if not (args.no_fused or args.no_recompute) else 8)
        except Exception as e:  # the cost model's defaults then decide (all ranks fail alike or the job ends)
            calib = {"error": repr(e)}
    t_round = calib["t_round_us"] * 1e-6 if calib and "t_round_us" in calib else None
    gather_bw = calib["gather_gbps"] * 1e9 if calib and "gather_gbps" in calib else None

    def make_engine(tune):
        e = SlabMultigrid(gny, gnx, n_para, h, 0.025, world=world, rank=rank, coarse_ordering=args.ordering,
                          use_graph=not args.no_graph, fused=not args.no_fused,
                          eager_dead_levels=not args.lazy_dead_levels, batch=args.batch or None,
                          recompute_presmoothing=not args.no_recompute, jacobi_omega=args.jacobi_omega,
                          tune_placement=tune, dist_levels=args.dist_levels or None, t_round=t_round, gather_bw=gather_bw,
                          overlap_min_rows=args.overlap_min_rows, graph_comm=args.graph_comm)
        fill_rhs(e, torch, n, rank)
        torch.cuda.synchronize()
        return e

    # What the same cycle costs on arrays taken as the allocator hands them out (no placement selection): a few
    # cycles on an engine of its own, built first -- as a process that does not select would build it -- and
    # released again.  One rank only (a rehearsal, outside the timed region of the benchmark proper).
    untuned_ms = None
    if world == 1 and not args.no_tune_placement and not args.batch and not args.no_untuned_leg and not args.no_fused:
        e0 = make_engine(False)
        e0.presmooth()
        e0.cycles(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.cycles(5)
        torch.cuda.synchronize()
        untuned_ms = 1e3 * (time.perf_counter() - t0) / 5
        e0.close()
        del e0
        torch.cuda.empty_cache()
    eng = make_engine(not args.no_tune_placement)
    nprob = args.batch or 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.presmooth()
    errs = eng.cycles(args.warmup) if args.warmup else []
    barrier()
    t0 = time.perf_counter()
    errs += eng.cycles(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    dt_local = dt
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- host time to issue one cycle: the same cycles again, the clock stopped before the device is waited for
    barrier()
    t0 = time.perf_counter()
    eng.ops.begin()
    for _ in range(args.steps):
        eng._run_cycle()
        eng._reduce_err()
    host_issue_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    eng._flush_pending()
    eng.ops.end()
    eng.materialize_residual()
    barrier()

    # ---- dominant kernel, timed with HIP events on the stream it is launched on.
    # With fused passes that is the finest level's single smoothing pass of the cycle
    # (prolongation + 6 Jacobi sweeps + residual + max-norm in one launch); without, the Jacobi sweep.
    L0, L1 = eng.plan.levels[0], eng.plan.levels[1]
    ops = eng.ops
    xs = [eng._xa[0], eng._xb[0]]
    iters = args.roofline_iters
    pts_launch = (L0.t1 - L0.t0) * (L0.nx - 2) * (args.batch or 1)

    def timed(fn):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.begin()
        for _ in range(2):
            fn(0)
        ev0.record(ops.stream)
        for k in range(iters):
            fn(k)
        ev1.record(ops.stream)
        ops.stream.synchronize()
        return ev0.elapsed_time(ev1) / iters

    sweep_ms = timed(lambda k: ops.jacobi(eng.b[0], xs[k & 1], xs[1 - (k & 1)], L0.nx, L0.t0, L0.t1, L0.h, True))
    single = {"kernel": "mg_stencil5_kernel<0 (Jacobi), 2, 1, streaming stores when the array is >= 256 MiB>", "avg_launch_ms": sweep_ms,
              "bytes_per_launch": 24 * pts_launch, "achieved": 24 * pts_launch / (sweep_ms * 1e-3) / 1e9}
    single["frac"] = single["achieved"] / HBM_PEAK_GBPS

    # what a plain device-to-device copy of one finest-level array reaches on this device, right now
    def plain_copy(k):
        with ops.ctx():
            xs[1 - (k & 1)].copy_(xs[k & 1])
    copy_ms = timed(plain_copy)
    copy_bytes = 2 * xs[0].numel() * 8
    device_copy = {"what": "tensor.copy_ of one finest-level x array (read + write)", "bytes": copy_bytes, "avg_ms": copy_ms,
                   "gbps": copy_bytes / (copy_ms * 1e-3) / 1e9}
    if eng.fused:
        # the correction and the restriction target as the cycle passes them: slab arrays of level 1, or, when
        # level 1 is replicated on every rank, this rank's window of it and the slab that is all-gathered
        rep1 = world > 1 and eng.n_dist == 1
        crs = eng._xa[1] if n_para > 1 else eng._xw0_pad
        if rep1:
            crs = eng._window(crs, 1)
        cb1 = eng._bslab if rep1 else eng.b[1]
        # exactly the launch the cycle makes on the finest level: prolongation + 6 sweeps + residual + max-norm
        # + full-weighting restriction of that residual into the next level's right-hand side (the residual
        # array itself is written once, when cycles() ends -- inside the timed region above).  The timed
        # launches overwrite b_1 with what the next cycle would compute anyway; x ping-pongs.
        pass_ms = timed(lambda k: ops.smooth_fused(eng.b[0], xs[k & 1], xs[1 - (k & 1)], None, crs, L0.nx, L1.nx,
                                                   L0.t0, L0.t1, L0.fix_lo, L0.fix_hi, L0.oj, 6, L0.h, eng.norm,
                                                   cb1, L1.nx, L1.t0, L1.t1, omega=args.jacobi_omega))
        streams = L0.ny * L0.ld * 8 * (args.batch or 1) >= (256 << 20)  # the instantiation with streaming stores
        dom_name = (f"mg_fused_kernel<6, true, false, true, true, 1, 0, {'false' if args.jacobi_omega == 1.0 else 'true'}, "
                    f"{'true' if streams else 'false'}> (finest level: prolongation + 6 Jacobi sweeps + residual + max-norm + restriction)")
        dom_ms = pass_ms
        alg_bpp = 18 + 6 * 24 + 24 + 10  # per-operation compulsory bytes x the operations this launch performs
        own_bpp = 8 + 2 + 8 + 8 + 2      # what the pass itself must move: R x, R coarse x, R b, W x, W coarse b
        tkey = "fused_finest_bytes_per_launch"
    else:
        dom_name, dom_ms, alg_bpp, own_bpp, tkey = single["kernel"], sweep_ms, 24, 24, "jacobi_finest_bytes_per_launch"
    own_bytes_launch = own_bpp * pts_launch
    alg_bytes_launch = alg_bpp * pts_launch
    achieved = own_bytes_launch / (dom_ms * 1e-3) / 1e9

    # ---- per-rank record, gathered to rank 0 (N > 1: what the first multi-GPU run needs to be read and re-tuned)
    mine = {"rank": rank, "device": dev_index, "ms_per_step_local": 1e3 * dt_local / args.steps, "host_issue_ms_per_cycle": host_issue_ms,
            "finest_pass_ms": dom_ms, "jacobi_sweep_ms": sweep_ms, "rounds_per_cycle": eng.rounds_per_cycle(),
            "passes_cut": eng.split_passes, "comm_mode": getattr(eng.comm, "mode", None),
            "placement_ms_chosen": (eng.placement or {}).get("ms_chosen_combination")}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    pcie = None
    if rank == 0 and not args.no_pcie and not args.batch:
        try:
            pcie = pcie_copy(torch, eng)
        except Exception as e:
            pcie = {"error": repr(e)}

    if rank == 0:
        unknowns = (gny - 2) * (gnx - 2) * nprob
        ms = 1e3 * dt / args.steps
        # HBM bytes per launch of the dominant kernel from rocprofv3 counters: measured on tools/pmc_probe.py at
        # 32769 x 32769 and kept in profiles/traffic.json with a digest of the kernel sources it was measured on
        traffic, traffic_source, cycle_traffic = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if n != 14 or args.batch or world != 1:
            traffic_source = "not measured for this workload (profiles/traffic.json holds the 32769 x 32769 launches)"
        elif not os.path.exists(tpath):
            traffic_source = "profiles/traffic.json is missing"
        else:
            try:
                from multigrid_amd._lib import sources_digest
                tj = json.load(open(tpath))
                if tj.get("csrc_sha256") == sources_digest():
                    traffic = tj.get(tkey)
                    if eng.fused and eng.eager_dead_levels and not args.no_recompute and args.jacobi_omega == 1.0 and args.ordering == "lexicographic":
                        cycle_traffic = (tj.get("cycle") or {}).get("hbm_bytes")
                    traffic_source = ("static: profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on tools/pmc_probe.py, "
                                      "same kernel sources, sha256 " + tj["csrc_sha256"][:12] + "); not collected in this run")
                else:
                    traffic_source = "profiles/traffic.json was measured on other kernel sources: not reported"
            except Exception as e:
                traffic_source = f"profiles/traffic.json unreadable: {e!r}"
        bcyc = algorithmic_bytes_per_cycle(eng.plan) * nprob
        recomputed = [l for l in range(1, n_para) if eng.fused and eng._recomputed(l)]
        own_cyc = (own_bytes_per_cycle(eng.plan, recomputed) if eng.fused else algorithmic_bytes_per_cycle(eng.plan)) * nprob
        out = {
            "metric": "unknowns*V-cycles/s", "value": unknowns * args.steps / dt, "unit": "unknowns*V-cycles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"{gny}x{gnx} fp64 Poisson V-cycle ({(gny - 1) // world + 1}x{gnx} per GPU), test.py 8-vortex RHS" if not args.batch
                             else f"{args.batch} independent {gny}x{gnx} fp64 Poisson V-cycles in one batch, test.py 8-vortex RHS"),
                "n": n, "n_para": n_para, "grid": [gny, gnx], "batch": args.batch or None, "unknowns": unknowns,
                "coarse_ordering": args.ordering, "jacobi_omega": args.jacobi_omega, "whole_domain_level": [eng.plan.whole[0].ny, eng.plan.whole[0].nx],
                "parallelism": f"row-slab x{world}", "hipgraph": bool(eng.use_graph), "fused_passes": bool(eng.fused), "recompute_presmoothing": bool(eng.recompute_presmoothing and eng.fused),
                "levels_with_recomputed_presmoothing": recomputed,
                "placement": eng.placement, "placement_sampling_seconds": round(eng.placement_seconds, 2),
                "halo_rows": eng.plan.halo, "distributed_levels": eng.n_dist if world > 1 else None,
                "exchange_overlap": ({"min_rows_per_rank": eng.overlap_min_rows, "passes_cut": eng.split_passes,
                                      "rounds_per_cycle": eng.rounds_per_cycle()} if world > 1 else None),
                "dead_whole_domain_levels": "computed every cycle" if eng.eager_dead_levels else "on demand", "backend": args.backend if world > 1 else None,
                "ranks_seen": dist.get_world_size() if world > 1 else 1,
                "comm": ({"mode": getattr(eng.comm, "mode", None), "calibration": calib, "cycle_is_one_graph": bool(eng.graph_comm),
                          "note": "mode native = raw RCCL send/recv groups issued by libmg_hip.so on the engine's streams; "
                                  "calibration = one exchange round and one all-gather timed before the engine was built, "
                                  "which the choice of distributed_levels was made from (--dist-levels / MG_T_ROUND_US / "
                                  "MG_GATHER_GBPS override)"} if world > 1 else None),
            },
            "host_issue_ms_per_cycle": host_issue_ms,
            "per_rank": per_rank,
            # not part of `value` (the timed cycles run on arrays resident in HBM): what moving one array costs
            "pcie": pcie,
            # the cycle on arrays as the allocator hands them out (no placement selection), 5 cycles on an engine of its own
            "untuned_ms_per_step": untuned_ms,
            "roofline": {
                "kernel": dom_name, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                "bytes_per_launch": own_bytes_launch, "bytes_per_point": own_bpp, "points_per_launch": pts_launch,
                "avg_launch_ms": dom_ms, "launches_timed": iters,
                "note": "achieved = the bytes the launch itself must move (every array element it reads or writes, once) / "
                        "its average duration by HIP events",
                "algorithmic_equivalent": {
                    "bytes_per_launch": alg_bytes_launch, "bytes_per_point": alg_bpp,
                    "gbps_equivalent": alg_bytes_launch / (dom_ms * 1e-3) / 1e9,
                    "note": "the operations of the launch priced one by one (prolongation 18, 6 sweeps x 24, residual 24, "
                            "restriction 10 B per point): a throughput equivalent for comparison with unfused code, not a bandwidth",
                },
                "single_jacobi_sweep": single,
                # the same achieved rate against a measured ceiling of the device instead of the specification's
                "device_copy": device_copy, "achieved_over_device_copy": achieved / device_copy["gbps"],
            },
            "cycle": {
                "own_bytes": own_cyc, "gbps_per_gpu": own_cyc / world / (ms * 1e-3) / 1e9,
                "frac_of_peak": own_cyc / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                # counters summed over every kernel of one cycle (same source and conditions as roofline.traffic)
                "traffic": cycle_traffic, "traffic_over_own_bytes": (cycle_traffic / own_cyc) if cycle_traffic else None,
                "algorithmic_equivalent_bytes": bcyc, "algorithmic_equivalent_gbps_per_gpu": bcyc / world / (ms * 1e-3) / 1e9,
                "max_abs_r_after_last_cycle": errs[-1] if errs else None,
            },
        }
        if world == 1 and not args.no_cpu_baseline and not args.batch:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
            except Exception as e:  # the GPU numbers stand on their own
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        import faulthandler
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
