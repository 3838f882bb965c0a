"""Rank-to-rank data movement of the slab engine over torch.distributed.

`SlabComm` wraps the three patterns the V-cycle needs -- neighbour row exchange, all-gather of the
first replicated level's right-hand side, max-reduction of the residual norm -- for two situations:

* direct: the tensors can be handed to the backend as they are (CUDA tensors on `nccl`, which is
  RCCL on ROCm and moves the rows over xGMI; or CPU tensors on `gloo` in the host-only tests);
* staged: CUDA tensors on a `gloo` group.  Every message goes through a host copy.  Slow, but it
  lets several ranks share ONE GPU, which is how the multi-rank path is rehearsed on a single-GPU
  machine (RCCL refuses two ranks on one device).
"""
import torch
import torch.distributed as dist


class SlabComm:
    def __init__(self, ops, group=None, world=1, rank=0):
        self.ops, self.group, self.world, self.rank = ops, group, world, rank
        self.staged = False
        self.rounds = 0  # neighbour exchanges + all-gathers issued so far (diagnostics)
        if world > 1:
            backend = dist.get_backend(group)
            self.staged = backend == "gloo" and ops.name == "hip"
            # one collective up front: every rank joins the communicator before the first
            # point-to-point batch (the backends create their channels lazily)
            t = ops.zeros(1, 1)
            self.allreduce_max(t.view(-1))

    def _peer(self, r):
        return r if self.group is None else dist.get_global_rank(self.group, r)

    def prepare(self, sends, recvs):
        """sends/recvs: lists of (1-D tensor view, group rank).  Returns a batch that run() can issue any
        number of times (the views stay bound to their arrays)."""
        if self.staged:
            return ("staged", list(sends), list(recvs))
        ops_ = [dist.P2POp(dist.isend, t, self._peer(p), self.group) for t, p in sends]
        ops_ += [dist.P2POp(dist.irecv, t, self._peer(p), self.group) for t, p in recvs]
        return ("direct", ops_)

    def run(self, batch):
        """Issue all sends and receives of a prepared batch as one group."""
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
        with self.ops.ctx():
            if not self.staged:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                return t
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)
            return t
