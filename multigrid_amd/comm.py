"""Rank-to-rank data movement of the slab engine.

`SlabComm` wraps the three patterns the V-cycle needs -- neighbour row exchange, all-gather of the
first replicated level's right-hand side, max-reduction of the residual norm -- for three situations:

* native: CUDA tensors on an `nccl` group (RCCL on ROCm; the rows move over xGMI).  The process group's
  RCCL communicator is handed to libmg_hip.so (mg_comm_attach), and every exchange is one prepared
  group of raw ncclSend/ncclRecv calls enqueued by the library on the engine's own stream
  (mg_halo_exchange): set up once per (arrays, depth) like the reference's persistent requests
  (core/parallel_multigrid.py:77-81 of the reference checkout), a few microseconds of host time per
  round, in order with the kernels of that stream.  MG_COMM=torch in the environment selects the next
  mode instead;
* direct: the tensors are handed to torch.distributed as they are (CUDA tensors on `nccl` through
  torch's own stream and bookkeeping, or CPU tensors on `gloo` in the host-only tests);
* staged: CUDA tensors on a `gloo` group.  Every message goes through a host copy.  Slow, but it
  lets several ranks share ONE GPU, which is how the multi-rank path is rehearsed on a single-GPU
  machine (RCCL refuses two ranks on one device).
"""
import ctypes as C
import os
import warnings

import torch
import torch.distributed as dist


def loaded_rccl_path():
    """File of the RCCL library mapped into this process (the one torch.distributed uses), or None."""
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if os.path.basename(path).startswith("librccl.so"):
                    return path
    except OSError:
        pass
    return None


def torch_nccl_comm(group, device):
    """The ncclComm_t (as an integer) torch's `nccl` backend of `group` uses on `device`."""
    pg = group if group is not None else dist.distributed_c10d._get_default_group()
    return pg._get_backend(torch.device(device))._comm_ptr()


class SlabComm:
    def __init__(self, ops, group=None, world=1, rank=0):
        self.ops, self.group, self.world, self.rank = ops, group, world, rank
        self.staged = False
        self.native = None   # mg_comm handle of libmg_hip.so when the exchanges are raw RCCL calls of the library
        self.mode = "none"
        self._halos = []
        self.rounds = 0  # neighbour exchanges + all-gathers issued so far (diagnostics)
        if world > 1:
            backend = dist.get_backend(group)
            self.staged = backend == "gloo" and ops.name == "hip"
            self.mode = "staged" if self.staged else "direct"
            # one collective up front: every rank joins the communicator before the first
            # point-to-point batch (the backends create their channels lazily)
            t = ops.zeros(1, 1)
            self.allreduce_max(t.view(-1))
            if backend == "nccl" and ops.name == "hip" and os.environ.get("MG_COMM", "native") != "torch":
                self._attach(torch_nccl_comm(group, ops.device))

    def _attach(self, comm_ptr):
        """Hand an existing RCCL communicator to the library; on failure the torch.distributed path stays in use."""
        from . import _lib
        L = _lib.lib()
        h = C.c_void_p()
        path = loaded_rccl_path()
        if L.mg_comm_attach(C.c_void_p(comm_ptr), path.encode() if path else None, C.byref(h)):
            warnings.warn("mg_comm_attach failed (" + L.mg_last_error().decode() + "): exchanges go through torch.distributed")
            return
        self.native, self.mode, self._L = h, "native", L

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def _native_peer(self, r):
        """Rank numbers of the communicator are the group's own."""
        return r

    def prepare(self, sends, recvs):
        """sends/recvs: lists of (1-D tensor view, group rank).  Returns a batch that run() can issue any
        number of times (the views stay bound to their arrays)."""
        if self.native is not None:
            from ._lib import check
            def table(msgs):
                n = len(msgs)
                return ((C.c_void_p * n)(*[t.data_ptr() for t, _ in msgs]), (C.c_int64 * n)(*[t.numel() for t, _ in msgs]),
                        (C.c_int * n)(*[self._native_peer(p) for _, p in msgs]))
            for t, _ in list(sends) + list(recvs):
                assert t.is_contiguous() and t.dtype == torch.float64
            sp, sc, sr = table(sends)
            rp, rc, rr = table(recvs)
            h = C.c_void_p()
            check(self._L.mg_halo_prepare(self.native, len(sends), sp, sc, sr, len(recvs), rp, rc, rr, C.byref(h)))
            self._halos.append(h)
            return ("native", h, len(sends) + len(recvs), list(sends) + list(recvs))  # the views keep their arrays alive
        if self.staged:
            return ("staged", list(sends), list(recvs))
        ops_ = [dist.P2POp(dist.isend, t, self._peer(p), self.group) for t, p in sends]
        ops_ += [dist.P2POp(dist.irecv, t, self._peer(p), self.group) for t, p in recvs]
        return ("direct", ops_)

    def run(self, batch):
        """Issue all sends and receives of a prepared batch as one group."""
        if batch[0] == "native":
            if batch[2]:
                from ._lib import check
                self.rounds += 1
                check(self._L.mg_halo_exchange(batch[1], self.ops._s))
            return
        if len(batch[1]) == 0 and (batch[0] == "direct" or len(batch[2]) == 0):
            return
        self.rounds += 1
        with self.ops.ctx():
            if batch[0] == "direct":
                for w in dist.batch_isend_irecv(batch[1]):
                    w.wait()
                return
            _, sends, recvs = batch
            host_s = [(t.cpu(), p) for t, p in sends]  # .cpu() waits for the kernels that produced the rows
            host_r = [(torch.empty(t.shape, dtype=t.dtype), p, t) for t, p in recvs]
            ops_ = [dist.P2POp(dist.isend, h, self._peer(p), self.group) for h, p in host_s]
            ops_ += [dist.P2POp(dist.irecv, h, self._peer(p), self.group) for h, p, _ in host_r]
            for w in dist.batch_isend_irecv(ops_):
                w.wait()
            for h, _, t in host_r:
                t.copy_(h)

    def sendrecv(self, sends, recvs):
        self.run(self.prepare(sends, recvs))

    def allgather_rows(self, out, inp):
        """out (world*R rows) <- the R rows `inp` of every rank, in rank order; both are whole-row
        (contiguous) slices of equally wide arrays."""
        self.rounds += 1
        if self.native is not None:
            from ._lib import check
            assert out.is_contiguous() and inp.is_contiguous() and out.numel() == self.world * inp.numel()
            check(self._L.mg_allgather_rows(self.native, inp.data_ptr(), out.data_ptr(), inp.numel(), self.ops._s))
            return
        with self.ops.ctx():
            if not self.staged:
                if dist.get_backend(self.group) == "nccl":
                    dist.all_gather_into_tensor(out.view(-1), inp.view(-1), group=self.group)
                else:
                    R = inp.shape[0]
                    dist.all_gather([out[q * R : (q + 1) * R] for q in range(self.world)], inp, group=self.group)
                return
            h = inp.cpu()
            hp = [torch.empty_like(h) for _ in range(self.world)]
            dist.all_gather(hp, h, group=self.group)
            out.copy_(torch.cat(hp, 0))

    def allreduce_max(self, t):
        """In-place MAX over ranks of a small tensor (integers, or NaN-free floats); returns it."""
        if self.native is not None and t.dtype == torch.int64 and t.is_contiguous():
            # the accumulators hold bit patterns of |r|: non-negative as signed integers, so the unsigned MAX agrees
            from ._lib import check
            check(self._L.mg_allreduce_max(self.native, t.data_ptr(), t.numel(), self.ops._s))
            return t
        with self.ops.ctx():
            if not self.staged:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                return t
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
            return t

    def close(self):
        """Release the prepared exchanges and the library's handle of the communicator (which stays torch's)."""
        if self.native is not None:
            for h in self._halos:
                self._L.mg_halo_destroy(h)
            self._halos = []
            self._L.mg_comm_destroy(self.native)
            self.native = None
