#!/usr/bin/env python3
"""Does a hipGraph capture on one stream survive RCCL's watchdog thread looking at its work items meanwhile?

The multi-GPU engine captures the replicated middle of the cycle into a hipGraph right after an RCCL all-gather
on the same stream.  torch's ProcessGroupNCCL polls the end events of outstanding work from a watchdog thread.
This probe issues a collective on the engine's stream, begins a capture, keeps it open for `hold` seconds (kernels
are recorded, the host sleeps, the watchdog polls), and ends it.  Cases:
  own_stream              the capture runs on a stream that never waits for an RCCL event (HipOps.graph_begin)
  same_stream_sync_sleep  on the stream the collective was issued on, after waiting for the device and 0.3 s more
                          (the watchdog has retired the work by then)
  same_stream_sync        the same without the 0.3 s
  same_stream             the same without waiting at all
Measured (profiles/r02_rccl_capture_probe.txt): the first two survive; the last two end the process -- the watchdog's
hipEventQuery fails with hipErrorCapturedEvent ("operation not permitted on an event last recorded in a capturing
stream") and torch terminates.

    python tools/rccl_capture_probe.py [case ...]
"""
import ctypes
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    from multigrid_amd.ops import HipOps
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s.getsockname()[1])
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ops = HipOps()
    a = ops.zeros(4096, 4096)
    big = ops.zeros(8192, 8192)
    gathered = ops.zeros(4096, 4096)
    hold = 0.4
    from multigrid_amd._lib import check
    for case in (sys.argv[1:] or ["own_stream", "same_stream_sync_sleep", "same_stream_sync", "same_stream"]):
        ok, err = True, ""
        try:
            for rep in range(5):
                ops.begin()
                for _ in range(20):
                    ops.zero_(big)  # the collective's end event is still pending when the capture begins
                with ops.ctx():
                    dist.all_gather_into_tensor(gathered.view(-1), a.view(-1))
                if "sync" in case:
                    ops.stream.synchronize()
                if "sleep" in case:
                    time.sleep(0.3)
                if case == "own_stream":
                    ops.graph_begin()
                else:
                    check(ops.L.mg_graph_begin(ops.stream.cuda_stream))
                ops.zero_(a)
                time.sleep(hold)
                ops.zero_(a)
                if case == "own_stream":
                    g = ops.graph_end()
                else:
                    g = ctypes.c_void_p()
                    check(ops.L.mg_graph_end(ops.stream.cuda_stream, ctypes.byref(g)))
                ops.graph_launch(g)
                ops.stream.synchronize()
                ops.graph_destroy(g)
        except Exception as e:  # noqa: BLE001
            ok, err = False, repr(e)[:300]
        print(f"case {case}: {'captures survived' if ok else 'FAILED: ' + err}", flush=True)
        if not ok:
            break
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
