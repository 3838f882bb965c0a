#!/usr/bin/env python3
"""Where does a hipGraph that holds raw RCCL calls stop working?  One rank, the rank as its own neighbour, the
library's calls (mg_halo_exchange, mg_allgather_rows, mg_allreduce_max) on the group's communicator.  Stages, each
announced before it starts (a watchdog dumps every thread's stack and ends the process after 25 s without progress):

  fork     the exchange is captured on the communication stream, which joins the capture through an event of the
           capturing stream and is joined back (the way a cut pass issues it), between two kernels
  eager    replays of that graph alternate with an eager MAX-reduction on the same communicator
  two      a second graph with the same calls is captured; the two are launched in turn
  gather   a graph with an exchange on the capturing stream itself and the all-gather

    python tools/rccl_graph_probe2.py [stage ...]      (default: all, in this order)
"""
import faulthandler
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    stages = [a for a in sys.argv[1:]] or ["fork", "eager", "two", "gather"]
    faulthandler.dump_traceback_later(25, exit=True, file=sys.stdout)
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from multigrid_amd.comm import SlabComm, torch_nccl_comm
    from multigrid_amd.ops import HipOps
    ops = HipOps(torch.device("cuda", 0))
    comm = SlabComm(ops, None, 1, 0)
    comm.allreduce_max(ops.zeros(1, 1).view(-1))
    comm._attach(torch_nccl_comm(None, ops.device))
    assert comm.mode == "native"
    H, rows, ld = 12, 1024, 4112
    ops.begin()
    with ops.ctx():
        x = torch.arange((rows + 2 * H) * ld, dtype=torch.float64, device="cuda").view(rows + 2 * H, ld)
        inp = x[H:H + 64]
        gathered = torch.zeros(64, ld, dtype=torch.float64, device="cuda")
        slots = ops.norm_slots()
        scratch = torch.ones(1 << 20, dtype=torch.float64, device="cuda")
    batch = comm.prepare([(x[H:2 * H].view(-1), 0), (x[rows:rows + H].view(-1), 0)], [(x[rows + H:].view(-1), 0), (x[:H].view(-1), 0)])
    comm.run(batch)
    comm.allgather_rows(gathered, inp)
    comm.allreduce_max(slots)
    ops.stream.synchronize()
    print("eager calls ran", flush=True)

    def tick(msg):
        faulthandler.cancel_dump_traceback_later()
        faulthandler.dump_traceback_later(25, exit=True, file=sys.stdout)
        print(msg, flush=True)

    def forked_sequence():
        ops.zero_(scratch)
        ready = ops.mark()
        ops.zero_(scratch)
        with ops.comm_side(ready):
            comm.run(batch)
            ops.zero_(scratch[: 1 << 10])
        ops.comm_join()
        ops.zero_(scratch)

    graphs = []
    if "fork" in stages or "eager" in stages or "two" in stages:
        tick("fork: capturing")
        with ops.capture() as cap:
            forked_sequence()
        graphs.append(cap.graph)
        tick("fork: captured; replaying")
        for k in range(3):
            ops.graph_launch(graphs[0])
            ops.stream.synchronize()
            tick(f"fork: replay {k} done")
    if "eager" in stages:
        for k in range(3):
            ops.graph_launch(graphs[0])
            tick(f"eager: graph {k} launched")
            comm.allreduce_max(slots)
            tick(f"eager: reduction {k} enqueued")
            ops.stream.synchronize()
            tick(f"eager: round {k} done")
    if "two" in stages:
        tick("two: capturing the second graph")
        with ops.capture() as cap:
            forked_sequence()
        graphs.append(cap.graph)
        for k in range(4):
            ops.graph_launch(graphs[k & 1])
            ops.stream.synchronize()
            tick(f"two: launch {k} done")
    if "gather" in stages:
        tick("gather: capturing")
        with ops.capture() as cap:
            ops.zero_(scratch)
            comm.run(batch)
            ops.zero_(scratch)
            comm.allgather_rows(gathered, inp)
            ops.zero_(scratch)
        tick("gather: captured; replaying")
        for k in range(3):
            ops.graph_launch(cap.graph)
            comm.allreduce_max(slots)
            ops.stream.synchronize()
            tick(f"gather: replay {k} done")
        graphs.append(cap.graph)
    ops.end()
    for g in graphs:
        ops.graph_destroy(g)
    tick("all stages passed")
    faulthandler.cancel_dump_traceback_later()
    comm.close()
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
