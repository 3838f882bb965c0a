#!/usr/bin/env python3
"""Can the engine's RCCL calls be captured into a hipGraph and replayed?  One rank, the rank as its own neighbour
(RCCL refuses two ranks on one device): a kernel, the halo exchange batch (send/receive of row blocks), a kernel, the
all-gather and the MAX-reduction are captured on the engine's stream with the library's own capture calls, replayed,
and compared with the eager sequence; host time per eager round against GPU time per captured round.

    python tools/rccl_graph_probe.py [p2p,gather,reduce] [--native]
        which of the three to put into the graph (default all); --native: the calls are the raw RCCL calls libmg_hip.so
        makes on the engine's stream (mg_halo_exchange, mg_allgather_rows, mg_allreduce_max) instead of torch.distributed's
"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import faulthandler
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from multigrid_amd.comm import SlabComm
    from multigrid_amd.ops import HipOps
    ops = HipOps(torch.device("cuda", 0))
    comm = SlabComm(ops, None, 1, 0)
    native = "--native" in sys.argv
    if native:
        from multigrid_amd.comm import torch_nccl_comm
        comm.allreduce_max(ops.zeros(1, 1).view(-1))  # the communicator exists once it has carried a collective
        comm._attach(torch_nccl_comm(None, ops.device))
        assert comm.mode == "native"
    print("communication calls:", comm.mode if native else "torch.distributed", flush=True)
    H, rows, ld = 12, 4096, 32784
    ops.begin()
    with ops.ctx():
        x = torch.arange((rows + 2 * H) * ld, dtype=torch.float64, device="cuda").view(rows + 2 * H, ld)
        x0 = x.clone()
        inp = x[H:H + 64]
        gathered = torch.zeros(64, ld, dtype=torch.float64, device="cuda")
        slots = ops.norm_slots()
    sends = [(x[H:2 * H].view(-1), 0), (x[rows:rows + H].view(-1), 0)]
    recvs = [(x[rows + H:].view(-1), 0), (x[:H].view(-1), 0)]
    batch = comm.prepare(sends, recvs)

    plain = [a for a in sys.argv[1:] if not a.startswith("--")]
    which = plain[0].split(",") if plain else ["p2p", "gather", "reduce"]

    def sequence():
        with ops.ctx():
            x.add_(1.0)  # the rows that go out change with every run
        if "p2p" in which:
            comm.run(batch)
        with ops.ctx():
            x[H:rows + H].mul_(0.5)
        if "gather" in which:
            comm.allgather_rows(gathered, inp)
        if "reduce" in which:
            comm.allreduce_max(slots)

    sequence()  # eager once: communicator and channels exist before the capture
    print("eager sequence ran", flush=True)
    ops.stream.synchronize()
    with ops.ctx():
        x.copy_(x0)
    for _ in range(3):
        sequence()
    ops.stream.synchronize()
    want, want_g = x.clone(), gathered.clone()
    # host time of an eager round
    t0 = time.perf_counter()
    for _ in range(50):
        comm.run(batch)
    host = (time.perf_counter() - t0) / 50
    ops.stream.synchronize()
    with ops.ctx():
        x.copy_(x0)
    ops.stream.synchronize()
    print(f"eager: {host * 1e6:.0f} us of host time per exchange round", flush=True)
    try:
        ops.graph_begin()
        print("capture begun", flush=True)
        sequence()
        print("sequence captured", flush=True)
        g = ops.graph_end()
    except Exception as e:  # noqa: BLE001
        print("capture failed:", repr(e))
        dist.destroy_process_group()
        return 1
    print("captured", flush=True)
    for _ in range(3):
        ops.graph_launch(g)
    ops.stream.synchronize()
    same = bool(torch.equal(x, want)) and bool(torch.equal(gathered, want_g))
    print("replay equals the eager sequence:", same, flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(ops.stream)
    for _ in range(50):
        ops.graph_launch(g)
    issue = (time.perf_counter() - t0) / 50
    e1.record(ops.stream)
    ops.stream.synchronize()
    print(f"eager: {host * 1e6:.0f} us of host time per exchange round; captured (exchange + all-gather + reduction + 2 kernels): "
          f"{issue * 1e6:.0f} us of host time and {e0.elapsed_time(e1) / 50 * 1e3:.0f} us of GPU time per replay", flush=True)
    ops.graph_destroy(g)
    ops.end()
    dist.destroy_process_group()
    return 0 if same else 2


if __name__ == "__main__":
    sys.exit(main())
