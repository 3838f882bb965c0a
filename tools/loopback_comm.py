"""A communicator for rehearsing ONE rank of an N-rank slab job alone on one GPU: SlabComm's interface for rank
`rank` of `world`, every peer being this process.  What the rank would send to a neighbour arrives in the halo it
would receive from the other neighbour (the rows are shaped alike), the all-gather fills every rank's rows with the
rank's own, the MAX-reduction sees one rank.  Values are therefore meaningless; the calls, their order, their sizes,
the streams they run on and their cost on the host and on the GPU are those of the real job.

rccl=True: the engine's own communication calls on a one-rank `nccl` group (which the caller has initialised),
every peer mapped to rank 0 -- native=True (default): the raw RCCL calls libmg_hip.so makes on the engine's streams
(mg_halo_exchange, mg_allgather_rows, mg_allreduce_max on the group's communicator); native=False: torch.distributed's.
rccl=False: plain device copies on the stream the exchange would run on.
Used by tools/rank_rehearsal.py and tests/test_gpu_rccl.py."""
import torch.distributed as dist

from multigrid_amd.comm import SlabComm, torch_nccl_comm


class LoopbackComm(SlabComm):
    def __init__(self, ops, world, rank, rccl=True, native=True):
        self.ops, self.group, self.world, self.rank = ops, None, world, rank
        self.staged, self.rounds = False, 0
        self.native, self._halos, self.mode = None, [], "copy"
        self.rccl = rccl
        if self.rccl:
            self.mode = "direct"
            t = ops.zeros(1, 1)
            self.allreduce_max(t.view(-1))
            if native:
                self._attach(torch_nccl_comm(None, ops.device))

    def _peer(self, r):
        return 0

    def _native_peer(self, r):
        return 0

    def prepare(self, sends, recvs):
        if self.rccl:
            n = min(len(sends), len(recvs))  # an end rank has one neighbour: equally many each way
            assert sorted(t.numel() for t, _ in sends[:n]) == sorted(t.numel() for t, _ in recvs[:n])
            return SlabComm.prepare(self, sends[:n], recvs[:n])
        pairs = []
        rs = list(recvs)
        for t, _ in sends:
            k = next((i for i, (u, _) in enumerate(rs) if u.numel() == t.numel()), None)
            if k is not None:
                pairs.append((rs.pop(k)[0], t))
        return ("copy", pairs)

    def run(self, batch):
        if batch[0] != "copy":
            return SlabComm.run(self, batch)
        if not batch[1]:
            return
        self.rounds += 1
        with self.ops.ctx():
            for dst, src in batch[1]:
                dst.copy_(src)

    def allgather_rows(self, out, inp):
        self.rounds += 1
        R = inp.shape[0]
        own = out[self.rank * R : (self.rank + 1) * R]
        if self.native is not None:  # a one-rank all-gather: the rank's rows into its place
            from multigrid_amd._lib import check
            check(self._L.mg_allgather_rows(self.native, inp.data_ptr(), own.data_ptr(), inp.numel(), self.ops._s))
        with self.ops.ctx():
            if self.native is not None:
                pass
            elif self.rccl:
                dist.all_gather_into_tensor(own.view(-1), inp.view(-1))
            else:
                own.copy_(inp)
            for q in range(self.world):  # the other ranks' rows: written once more on this device
                if q != self.rank:
                    out[q * R : (q + 1) * R].copy_(own)

    def allreduce_max(self, t):
        if self.native is not None:
            return SlabComm.allreduce_max(self, t)
        if self.rccl:
            with self.ops.ctx():
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t
