#!/usr/bin/env python3
"""One rank of an N-rank weak-scaling job, alone on one GPU, with the rank as its own neighbours.

The multi-GPU engine cuts the grid into row slabs; what a rank does per V-cycle -- its passes (cut into
interior and slab ends around the halo exchange), the exchange rounds, the all-gather of the first
replicated level's right-hand side, the replicated levels, the MAX-reduction of the norm -- does not depend
on what the neighbours' rows contain.  This tool builds rank `--rank` of `--world` ranks exactly as
`bench.py --gpus N` builds it (same slab, same number of distributed levels, same halo depth, same
placement selection) and replaces the communicator by one that moves the same rows between the same
buffers ON THIS DEVICE:

  --comm rccl   (default) the engine's own torch.distributed calls on a one-rank `nccl` group, every peer
                mapped to rank 0: RCCL's host time per round and its copy kernels are the real ones, the
                bytes cross HBM instead of xGMI;
  --comm copy   plain device copies on the stream the exchange would run on (no RCCL in the process).

Numbers are meaningless (the halo rows hold the rank's own rows); times are not: the cycle time here is
what one rank's GPU and host need when the links cost nothing, i.e. t(1 rank) / t(this) is the ceiling of
the weak-scaling efficiency that the slab engine's own overheads allow.  Prints one JSON line.

    python tools/rank_rehearsal.py --world 8 --rank 3 --steps 10
    rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/rank_rehearsal.py --world 8 --rank 3
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=None, help="default: a middle rank (two neighbours)")
    ap.add_argument("--exp", dest="n", type=int, default=14)
    ap.add_argument("--jacobi-levels", dest="n_para", type=int, default=0)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--comm", default="rccl", choices=["rccl", "rccl-torch", "copy"],
                    help="rccl: raw RCCL calls made by libmg_hip.so on the engine's streams (the default of the N > 1 engine); "
                         "rccl-torch: the same exchanges through torch.distributed; copy: device copies, no RCCL")
    ap.add_argument("--no-tune-placement", action="store_true")
    ap.add_argument("--solve-order", action="store_true",
                    help="read every cycle's norm before the next cycle is issued, as solve() must")
    ap.add_argument("--overlap-min-rows", type=int, default=4096)
    ap.add_argument("--dist-min-rows", type=int, default=None)
    ap.add_argument("--strong", action="store_true", help="strong scaling: the (2**(n+1)+1)^2 grid cut into N slabs (bench.py --strong)")
    ap.add_argument("--rccl-high-priority", action="store_true", help="RCCL's own streams at high priority")
    ap.add_argument("--engine-args", default="{}", help="JSON dict of further SlabMultigrid keyword arguments")
    ap.add_argument("--native", action="store_true",
                    help="the C handle of the several-rank solver (mg_dist_*, MG_DIST_LOOPBACK) in the place of the Python engine: "
                         "a rank's whole cycle, RCCL calls included, is one graph launch issued by the library")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from loopback_comm import LoopbackComm
    from multigrid_amd.engine import SlabMultigrid
    from multigrid_amd.levels import choose_n_para
    from multigrid_amd.ops import HipOps

    world = args.world
    rank = world // 2 if args.rank is None else args.rank
    torch.cuda.set_device(0)
    if args.comm != "copy":
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True) if args.rccl_high_priority else None
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=opts)

    n = args.n
    gnx = 2 ** (n + 1) + 1
    gny = (1 if args.strong else world) * 2 ** (n + 1) + 1
    n_para = args.n_para or choose_n_para(gny, gnx, world, at_least=max(1, n - 4))
    h = 20.0 / 2 ** (n + 1)
    if args.native:
        return native(args, world, rank, gny, gnx, n_para, h, torch, dist)
    ops = HipOps()
    comm = LoopbackComm(ops, world, rank, rccl=args.comm != "copy", native=args.comm == "rccl") if world > 1 else None
    extra = json.loads(args.engine_args)
    eng = SlabMultigrid(gny, gnx, n_para, h, 0.025, ops=ops, world=world, rank=rank, comm=comm, eager_dead_levels=True,
                        tune_placement=not args.no_tune_placement, overlap_min_rows=args.overlap_min_rows,
                        dist_min_rows=args.dist_min_rows, **extra)
    import bench
    bench.fill_rhs(eng, torch, n, rank)
    torch.cuda.synchronize()

    def run(k):
        if not args.solve_order:
            return eng.cycles(k)
        errs = []
        eng.ops.begin()
        for _ in range(k):
            eng._run_cycle()
            errs.append(eng._global_err())
        eng._flush_pending()
        eng.ops.end()
        return errs

    eng.presmooth()
    run(args.warmup)
    torch.cuda.synchronize()
    r0 = comm.rounds if comm else 0
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rounds = ((comm.rounds - r0) / args.steps) if comm else 0
    # host time to issue one cycle: the same cycles again, the clock stopped before the device is waited for
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.ops.begin()
    for _ in range(args.steps):
        eng._run_cycle()
        eng._reduce_err()
    t_issue = time.perf_counter() - t0
    eng._flush_pending()
    eng.ops.end()
    torch.cuda.synchronize()
    out = {"world": world, "rank": rank, "comm": args.comm, "comm_mode": comm.mode if comm else None, "scaling": "strong" if args.strong else "weak", "grid": [gny, gnx], "n_para": n_para,
           "distributed_levels": eng.n_dist if world > 1 else None, "halo_rows": eng.plan.halo,
           "ms_per_cycle": 1e3 * dt / args.steps, "order": "solve" if args.solve_order else "cycles",
           "host_issue_ms_per_cycle": 1e3 * t_issue / args.steps,
           "rounds_per_cycle": rounds,
           "passes_cut": eng.split_passes, "placement_seconds": round(eng.placement_seconds, 2),
           "placement": eng.placement}
    print(json.dumps(out), flush=True)
    eng.close()
    if args.comm != "copy":
        dist.destroy_process_group()


def native(args, world, rank, gny, gnx, n_para, h, torch, dist):
    import ctypes as C

    import bench
    from multigrid_amd import _lib
    from multigrid_amd.comm import loaded_rccl_path, torch_nccl_comm
    from multigrid_amd.engine import SlabMultigrid
    from multigrid_amd.solver import NativeDistSolver
    assert args.comm == "rccl" and world > 1, "--native rehearses the library's RCCL calls on a one-rank nccl group"
    t = torch.zeros(1, device="cuda")
    dist.all_reduce(t)
    c = C.c_void_p()
    path = loaded_rccl_path()
    _lib.check(_lib.lib().mg_comm_attach(C.c_void_p(torch_nccl_comm(None, torch.device("cuda", 0))), path.encode() if path else None, C.byref(c)))
    s = NativeDistSolver(gny, gnx, n_para, h, c, rank, world, loopback=True)
    info = s.info()
    # the rank's rows of the bench's right-hand side, made by the engine's own generator on a throw-away slab
    L = SlabMultigrid(gny, gnx, n_para, h, 0.025, world=world, rank=rank, comm=type("C", (), {"staged": True, "mode": "none"})(),
                      tune_placement=False, dist_levels=info["n_dist"])
    bench.fill_rhs(L, torch, args.n, rank)
    s.upload("b", L.local_rows(L.b0))
    del L
    torch.cuda.empty_cache()
    s.presmooth()
    s.cycles(args.warmup)
    torch.cuda.synchronize()
    r0 = s.info()["rounds"]
    t0 = time.perf_counter()
    s.cycles(args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"world": world, "rank": rank, "comm": "rccl", "engine": "C handle (mg_dist_*), loopback", "scaling": "strong" if args.strong else "weak",
           "grid": [gny, gnx], "n_para": n_para, "distributed_levels": info["n_dist"], "halo_rows": 12,
           "ms_per_cycle": 1e3 * dt / args.steps, "order": "cycles", "rounds_per_cycle": (s.info()["rounds"] - r0) / args.steps}
    # host time to issue a cycle: the library's own loop returns when the last cycle's norm has arrived, so the time the
    # host is busy is measured on single-cycle calls minus the device's time for them
    print(json.dumps(out), flush=True)
    s.close()
    _lib.lib().mg_comm_destroy(c)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
