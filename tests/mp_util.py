"""Running a function on several spawned ranks from a test: results come back through a queue, a rank that dies
fails the test at once, and no process outlives the test."""
import queue
import socket
import time


def rank_watchdog(seconds=150):
    """Call first in a spawned rank: if the rank is still running after `seconds`, every thread's stack goes to
    stderr and the process exits, so that a deadlock among ranks fails its test with the place each rank waits at
    instead of hanging until the test's own time limit."""
    import faulthandler
    import sys
    faulthandler.dump_traceback_later(seconds, exit=True, file=sys.stderr)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_ranks(target, world, args, timeout=280):
    """Start target(rank, world, port, *args, q) on `world` spawned processes; each must put one tuple whose first
    element is its rank into q.  Returns {rank: rest of the tuple}.  Raises as soon as a process exits with a
    non-zero code or `timeout` seconds pass; whatever is still alive is terminated either way."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(rk, world, port) + tuple(args) + (q,)) for rk in range(world)]
    for p in procs:
        p.start()
    out, deadline = {}, time.monotonic() + timeout
    try:
        while len(out) < world:
            try:
                msg = q.get(timeout=1.0)
                out[msg[0]] = msg[1:]
                continue
            except queue.Empty:
                pass
            dead = [(k, p.exitcode) for k, p in enumerate(procs) if p.exitcode not in (None, 0)]
            if dead:
                raise AssertionError(f"ranks exited with (rank, code) {dead} before reporting")
            if time.monotonic() > deadline:
                raise AssertionError(f"ranks {sorted(set(range(world)) - set(out))} did not report within {timeout} s")
        for p in procs:
            p.join(timeout=60)
        codes = [p.exitcode for p in procs]
        assert codes == [0] * world, f"exit codes {codes}"
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    return out
