"""The four-thread communicator of multigrid_amd.spmd on its own (no GPU): the collectives a per-rank driver of the
reference uses on MPI.COMM_WORLD (test/test.py:34-35,83; core/parallel_multigrid.py:132-133,239,342-345 of the
reference checkout) behave as mpi4py's do for four ranks."""
import threading

import numpy as np
import pytest

from multigrid_amd import spmd


class _World:
    device = None


def _on_four_threads(fn):
    comm = spmd.ThreadComm(_World())
    out, errs = [None] * 4, [None] * 4

    def body(r):
        spmd._tls.ctx = (comm, r)
        try:
            out[r] = fn(comm, r)
        except BaseException as e:  # noqa: BLE001
            errs[r] = e
            comm._barrier.abort()
        finally:
            spmd._tls.ctx = None

    ts = [threading.Thread(target=body, args=(r,)) for r in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    return out, errs


def test_collectives_of_the_thread_communicator():
    def fn(comm, r):
        assert comm.Get_rank() == r == comm.rank and comm.Get_size() == 4
        total = comm.allreduce(r + 1)                    # default op: SUM (the reference's stop vote, :342-345)
        worst = comm.reduce(float(r) * 0.5, spmd.MPI.MAX, 0)  # test/test.py:83
        least = comm.allreduce(r, spmd.MPI.MIN)
        word = comm.bcast("from two" if r == 2 else None, 2)
        buf = np.full((3, 2), float(r))
        comm.Bcast(buf, root=1)                          # gather_blocks' broadcasts, :132-133
        both = comm.Bcast([buf, None], root=3)
        every = comm.allgather(r * r)
        root_only = comm.gather(r, 1)
        comm.Barrier()
        computed = comm.on_rank0(lambda: ["a", "b", "c", "d"])[r]
        return total, worst, least, word, buf.copy(), both, every, root_only, computed

    out, errs = _on_four_threads(fn)
    assert errs == [None] * 4
    for r, (total, worst, least, word, buf, both, every, root_only, computed) in enumerate(out):
        assert total == 10 and least == 0 and word == "from two" and every == [0, 1, 4, 9]
        assert worst == (1.5 if r == 0 else None)
        assert np.array_equal(buf, np.full((3, 2), 1.0))
        assert root_only == ([0, 1, 2, 3] if r == 1 else None)
        assert computed == "abcd"[r]


def test_an_error_on_rank0_reaches_every_rank_and_a_lost_rank_frees_the_others():
    def fn(comm, r):
        def boom():
            raise KeyError("rank 0's work failed")
        comm.on_rank0(boom)

    out, errs = _on_four_threads(fn)
    assert all(isinstance(e, KeyError) for e in errs)

    def fn2(comm, r):
        if r == 3:
            raise ValueError("gone")
        comm.Barrier()

    out, errs = _on_four_threads(fn2)
    assert isinstance(errs[3], ValueError) and all(isinstance(e, RuntimeError) for e in errs[:3])


def test_comm_world_exists_inside_run_only():
    with pytest.raises(RuntimeError):
        spmd.MPI.COMM_WORLD
    assert spmd.current() is None
