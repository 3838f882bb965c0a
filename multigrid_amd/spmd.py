"""Running a per-rank (SPMD) driver of the reference unchanged, in one process.

The reference's drivers are written for `mpiexec -n 4`: every rank executes the same script, asks
`MPI.COMM_WORLD` for its rank, builds ONE `Multigrid(b, a, R, h, eps, n, n_para)` on its own block and calls
`solve()`, `comm.reduce(...)`, `gather_blocks(comm, ...)` together with the other three
(test/test.py:33-87, core/parallel_multigrid.py:351-437 of the reference checkout).  This module lets such a
driver run as it is, without mpi4py or mpiexec, with the four blocks advanced together on one GPU:

    from multigrid_amd.spmd import MPI, run            # in place of `from mpi4py import MPI`
    from multigrid_amd.parallel_multigrid import Multigrid, gather_blocks, split

    def driver():                                       # the reference's per-rank program, as written
        comm = MPI.COMM_WORLD
        rank = comm.Get_rank()
        ...
        poisson = Multigrid(b, a, R, h, eps, n, n_para)   # no rank= / world= keywords
        err = poisson.solve()
        err = comm.reduce(err, MPI.MAX, 0)
        gather_blocks(comm, a, a_full)

    results = run(driver)                               # four threads, one per rank; the four return values

`run` starts four threads.  Inside them `MPI.COMM_WORLD` is a communicator of four ranks whose collectives are
rendezvous among the threads (the surface the reference uses: Get_rank, Get_size, Barrier, reduce, allreduce, bcast,
Bcast; MAX, MIN, SUM), `Multigrid(...)` without keywords takes the calling thread's rank and the run's `BlockWorld`,
`solve()` is the collective solve of all four blocks (computed once, by the library's kernels, when the four ranks have
arrived), and `gather_blocks(comm, block, full)` gives every rank the stitched whole-domain array, as the reference's four
broadcasts do (core/parallel_multigrid.py:117-138).  Results equal the reference's four-rank runs bit for bit.
"""
import threading

__all__ = ["MPI", "run", "current"]

_tls = threading.local()


def current():
    """(communicator, rank) of the calling thread inside run(), else None."""
    return getattr(_tls, "ctx", None)


class _Op:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __repr__(self):
        return f"MPI.{self.name}"


class ThreadComm:
    """Four ranks, one thread each: the part of mpi4py's communicator interface the reference uses."""

    size = 4

    def __init__(self, world):
        self.world = world                       # the BlockWorld the ranks' Multigrid objects share
        self._barrier = threading.Barrier(self.size)
        self._slots = [None] * self.size
        self._result = None
        self.gpu_lock = threading.Lock()         # one rank at a time issues GPU work outside the collective solve

    # ---- identity
    def Get_rank(self):
        ctx = current()
        if ctx is None or ctx[0] is not self:
            raise RuntimeError("this communicator belongs to the ranks started by multigrid_amd.spmd.run")
        return ctx[1]

    def Get_size(self):
        return self.size

    rank = property(Get_rank)

    # ---- rendezvous
    def _wait(self):
        try:
            self._barrier.wait()
        except threading.BrokenBarrierError:
            raise RuntimeError("another rank left the collective (it raised, or the run was aborted)") from None

    def _exchange(self, value):
        """Every rank deposits a value; every rank gets the list of all four."""
        r = self.Get_rank()
        self._slots[r] = value
        self._wait()
        out = list(self._slots)
        self._wait()  # nobody overwrites a slot before everybody has read
        return out

    def on_rank0(self, fn):
        """All ranks arrive, rank 0 computes fn(), all ranks return its result."""
        r = self.Get_rank()
        self._wait()
        if r == 0:
            try:
                self._result = (True, fn())
            except BaseException as e:  # noqa: BLE001  -- handed to every rank
                self._result = (False, e)
        self._wait()
        ok, val = self._result
        try:
            self._wait()  # rank 0 does not start the next collective's work before everybody has read this one's
        except RuntimeError:
            if ok:
                raise
        if not ok:  # (a rank that has already raised this may have aborted the rendezvous: the error itself counts)
            raise val
        return val

    def Barrier(self):
        self._wait()

    barrier = Barrier

    # ---- collectives on Python objects (lower-case, pickle-path in mpi4py)
    def allreduce(self, value, op=None):
        vals = self._exchange(value)
        op = SUM if op is None else op
        acc = vals[0]
        for v in vals[1:]:
            acc = op.fn(acc, v)
        return acc

    def reduce(self, value, op=None, root=0):
        acc = self.allreduce(value, op)
        return acc if self.Get_rank() == root else None

    def bcast(self, obj=None, root=0):
        return self._exchange(obj)[root]

    def gather(self, obj, root=0):
        vals = self._exchange(obj)
        return vals if self.Get_rank() == root else None

    def allgather(self, obj):
        return self._exchange(obj)

    # ---- buffers (upper-case in mpi4py): numpy arrays, in place
    def Bcast(self, buf, root=0):
        import numpy as np
        arr = buf[0] if isinstance(buf, (list, tuple)) else buf
        src = self._exchange(arr)[root]
        if arr is not src:
            np.copyto(arr, src)
        self._wait()  # the root does not change its buffer while the others copy


MAX = _Op("MAX", lambda a, b: a if a >= b else b)
MIN = _Op("MIN", lambda a, b: a if a <= b else b)
SUM = _Op("SUM", lambda a, b: a + b)


class _MPI:
    """`from multigrid_amd.spmd import MPI` stands in for `from mpi4py import MPI` inside run()."""
    MAX, MIN, SUM = MAX, MIN, SUM

    @property
    def COMM_WORLD(self):
        ctx = current()
        if ctx is None:
            raise RuntimeError("MPI.COMM_WORLD exists inside multigrid_amd.spmd.run(driver) only: the four ranks are four "
                               "threads of this process")
        return ctx[0]


MPI = _MPI()


def run(driver, *args, world=None, **kwargs):
    """Execute driver(*args, **kwargs) on four ranks (threads) and return their four results, rank order.
    An exception in any rank aborts the others' pending collectives and is re-raised here."""
    import torch
    from .parallel_multigrid import BlockWorld
    world = BlockWorld() if world is None else world
    comm = ThreadComm(world)
    results, errors = [None] * 4, [None] * 4

    def body(rank):
        _tls.ctx = (comm, rank)
        try:
            torch.cuda.set_device(world.device)
            results[rank] = driver(*args, **kwargs)
        except BaseException as e:  # noqa: BLE001
            errors[rank] = e
            comm._barrier.abort()
        finally:
            _tls.ctx = None

    threads = [threading.Thread(target=body, args=(r,), name=f"mg-rank-{r}") for r in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    first = next((e for e in errors if e is not None and not (isinstance(e, RuntimeError) and "left the collective" in str(e))), None)
    first = first if first is not None else next((e for e in errors if e is not None), None)
    if first is not None:
        raise first
    return results
