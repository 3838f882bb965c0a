"""Row-slab V-cycle engine: the reference's ``Multigrid.solve()`` schedule on 1..N GPUs.

The schedule is the reference's (core/parallel_multigrid.py:232-348 of the reference checkout):
one Jacobi sweep + residual, then V-cycles of
  descent   restrict r, and on each coarser Jacobi level: x = 0, 3 Jacobi sweeps, residual, restrict;
  coarse    3 x (forward + backward SOR, omega 1.5) on the first whole-domain level, then the
            reference's further whole-domain levels (whose result its ascent loop never feeds back,
            :302 -- reproduced unless fix_coarse_ascent is set);
  ascent    prolong + add, 6 Jacobi sweeps per Jacobi level; residual + max-norm on the finest;
until max|r| <= epsilon, the divergence latch trips (:337-340) or 100 cycles have run (:270).

What differs from the reference is where the data lives and how it is cut:
* the four overlapping 2x2 blocks become P row slabs of the global grid (one per GPU/rank) with a
  one-row halo on each side; a slab boundary is exchanged as ONE contiguous row per neighbour;
* restriction/prolongation need no per-rank offsets table: with slab boundaries on even rows the
  fine centre of local coarse row T is local fine row 2T-1 on every rank (ofst_j = -1, ofst_i = 0);
* the whole-domain levels live on rank 0 only: their right-hand side is gathered there, the tail
  runs as one hipGraph, and the correction is scattered back (the reference replicates the tail on
  every rank behind four broadcasts, :117-138);
* Jacobi sweeps ping-pong between two arrays instead of copying a scratch array back;
* on one GPU the whole V-cycle is captured once into a hipGraph and replayed per cycle.

The engine is written against an ``ops`` object (multigrid_amd.ops.HipOps: the HIP kernels behind
the C ABI) and a torch.distributed process group for the exchanges.
"""
import torch
import torch.distributed as dist

from .comm import SlabComm
from .levels import make_plan


class SlabMultigrid:
    N1 = 3  # pre-smoothing sweeps  (parallel_multigrid.py:255)
    N2 = 6  # post-smoothing sweeps (parallel_multigrid.py:256)

    def __init__(self, gny, gnx, n_para, h0, epsilon, *, ops=None, group=None, world=None, rank=None,
                 coarse_ordering="lexicographic", fix_coarse_ascent=False, keep_ascent_residuals=False,
                 max_cycles=100, use_graph=True, fused=True):
        if ops is None:
            from .ops import HipOps
            ops = HipOps()
        self.ops = ops
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
            rank = dist.get_rank(group) if world > 1 else 0
        self.world, self.rank = world, rank
        self.comm = SlabComm(ops, group, world, rank)
        self.plan = make_plan(gny, gnx, n_para, world, rank, h0)
        self.n_para = n_para
        self.epsilon = epsilon
        self.coarse_ordering = coarse_ordering
        self.fix_coarse_ascent = fix_coarse_ascent
        self.keep_ascent_residuals = keep_ascent_residuals
        self.max_cycles = max_cycles
        # fused multi-sweep passes need halos as deep as the fused depth; until the slab exchange
        # moves that many rows they are used on one rank only (results are identical either way)
        self.fused = fused and world == 1 and getattr(ops, "supports_fused", False)
        self.use_graph = use_graph and getattr(ops, "supports_graphs", False)
        self.it = 0
        self.err_history = []
        self.diverged = False

        P = self.plan
        z = ops.zeros
        # Jacobi levels: two x arrays (ping-pong), b, r.  b has one more level (the staging level).
        self._xa = [z(L.ny, L.ld) for L in P.levels[:n_para]]
        self._xb = [z(L.ny, L.ld) for L in P.levels[:n_para]]
        self.b = [z(L.ny, L.ld) for L in P.levels]
        self.r = [z(L.ny, L.ld) for L in P.levels[:n_para]]
        self._cur0 = 0  # which of (_xa[0], _xb[0]) holds the finest x
        self.norm = ops.norm_slots()
        # whole-domain levels (rank 0).  Level 0 of them on one rank aliases the staging slab:
        # local row t of the slab is global row t-1, so the grid is the slab minus its first row.
        self.xw, self.bw, self.rw = [], [], []
        S = P.levels[n_para]
        if world == 1:
            self._xw0_slab = z(S.ny, S.ld)
            self.xw.append(self._xw0_slab[1:])
            self.bw.append(self.b[n_para][1:])
            self.rw.append(z(P.whole[0].ny, P.whole[0].ld))
        elif rank == 0:
            W = P.whole[0]
            self._xw0_pad = z(W.ny + 1, W.ld)  # row 0 is a dummy so that slab windows start at p*R
            self.xw.append(self._xw0_pad[1:])
            self.bw.append(z(W.ny, W.ld))
            self.rw.append(z(W.ny, W.ld))
        if rank == 0:
            for W in P.whole[1:]:
                self.xw.append(z(W.ny, W.ld))
                self.bw.append(z(W.ny, W.ld))
                self.rw.append(z(W.ny, W.ld))
        if world > 1:
            self._gather_buf = z(world * S.ny, S.ld) if rank == 0 else None
            self._scatter_buf = z(world * S.ny, S.ld) if rank == 0 else None
        self._graphs = {}
        self._tail_graph = None

    # ------------------------------------------------------------------ data access
    @property
    def x0(self):
        """Finest-level solution slab (rows x ld tensor; columns >= nx are padding)."""
        return (self._xa if self._cur0 == 0 else self._xb)[0]

    @property
    def b0(self):
        return self.b[0]

    @property
    def r0(self):
        return self.r[0]

    def owned_rows(self, level=0):
        """(first, last+1) local rows this rank owns, the global boundary rows included."""
        L = self.plan.levels[level]
        return 1, L.rows + (2 if self.rank == self.world - 1 else 1)

    def global_row_of_local(self, t, level=0):
        return self.rank * self.plan.levels[level].rows - 1 + t

    def set_x_rows(self, rows):
        """Set the owned rows of the finest x (both ping-pong arrays, so boundary values agree)."""
        a, e = self.owned_rows()
        for arr in (self._xa[0], self._xb[0]):
            arr[a:e, : self.plan.gnx] = rows

    def set_b_rows(self, rows):
        a, e = self.owned_rows()
        self.b[0][a:e, : self.plan.gnx] = rows

    def local_rows(self, arr):
        """The owned rows of a finest-level array as a (rows x gnx) view."""
        a, e = self.owned_rows()
        return arr[a:e, : self.plan.gnx]

    # ------------------------------------------------------------------ halo exchange
    def _exchange(self, arr, L, up=True, down=True):
        """Fill the halo rows of `arr` from the neighbouring ranks: row 0 <- last owned row of the
        rank above, row R+1 <- first owned row of the rank below.  `up`/`down` select which of my
        halos are needed (the transfer operators need only one)."""
        if self.world == 1:
            return
        R, nx = L.rows, L.nx
        sends, recvs = [], []
        if self.rank > 0:
            if down:  # the rank above needs my first owned row as its lower halo
                sends.append((arr[1, :nx], self.rank - 1))
            if up:
                recvs.append((arr[0, :nx], self.rank - 1))
        if self.rank < self.world - 1:
            if up:    # the rank below needs my last owned row as its upper halo
                sends.append((arr[R, :nx], self.rank + 1))
            if down:
                recvs.append((arr[R + 1, :nx], self.rank + 1))
        self.comm.sendrecv(sends, recvs)

    # ------------------------------------------------------------------ building blocks
    def _jacobi_sweeps(self, l, src, dst, n, finest=False):
        """n x (sweep + exchange), ping-ponging src -> dst -> src ...; returns the array holding the result."""
        L, o = self.plan.levels[l], self.ops
        for _ in range(n):
            o.jacobi(self.b[l], src, dst, L.nx, L.t0, L.t1, L.h, finest)
            self._exchange(dst, L)
            src, dst = dst, src
        return src

    @staticmethod
    def _chunks(n, most=6):
        """n sweeps as passes of at most `most` sweeps, as even as possible (8 -> 4+4)."""
        k = (n + most - 1) // most
        return [n // k + (1 if i < n % k else 0) for i in range(k)]

    def _smooth_fused(self, l, src, crs, cnx, n, r=None, norm=None):
        """[src += P(crs)], n Jacobi sweeps, [residual into r (+ norm)] on level l as fused passes that
        ping-pong between the level's two x arrays.  src None = start from zero.  Returns the array
        that holds the result."""
        L, o = self.plan.levels[l], self.ops
        xa, xb = self._xa[l], self._xb[l]
        chunks = self._chunks(n)
        for ci, k in enumerate(chunks):
            dst = xa if (src is None or src is xb) else xb
            last = ci == len(chunks) - 1
            if last and norm is not None:
                o.norm_reset(norm)
            o.smooth_fused(self.b[l], src, dst, r if last else None, crs if ci == 0 else None, L.nx, cnx, L.t0, L.t1,
                           L.t0 - 1, L.t1, -1, k, L.h, norm if last else None)
            src = dst
        return src

    def _smooth_level0(self, n, want_norm, crs=None):
        """smooth_parallel on the finest level: exchange, n sweeps, residual (:100-114); with `crs`
        the preceding prolongation (:322) is part of it."""
        L, o = self.plan.levels[0], self.ops
        xs = [self._xa[0], self._xb[0]]
        cur = xs[self._cur0]
        if self.fused:
            res = self._smooth_fused(0, cur, crs, self.plan.levels[1].nx, n, self.r[0], self.norm if want_norm else None)
            self._cur0 = 0 if res is xs[0] else 1
            return
        if crs is not None:
            o.prolong_add(crs, self.plan.levels[1].nx, cur, L.nx, L.t0, L.t1, 0, -1)
        self._exchange(cur, L)
        res = self._jacobi_sweeps(0, cur, xs[1 - self._cur0], n, finest=True)
        self._cur0 = 0 if res is xs[0] else 1
        if want_norm:
            o.norm_reset(self.norm)
        o.residual(self.r[0], res, self.b[0], L.nx, L.t0, L.t1, L.h, self.norm if want_norm else None, finest=True)

    def _tail(self):
        """Whole-domain levels on rank 0 (parallel_multigrid.py:291-304)."""
        o, W = self.ops, self.plan.whole
        nw = len(W)
        o.zero_(self.xw[0] if self.world > 1 else self._xw0_slab)
        o.sor(self.bw[0], self.xw[0], self.rw[0], W[0].nx, W[0].h, self.N1, self.coarse_ordering)
        for i in range(1, nw):
            o.zero_(self.xw[i])
            o.restrict(self.rw[i - 1], W[i - 1].nx, self.bw[i], W[i].nx, 1, W[i].ny - 1, 0, 0)
            o.sor(self.bw[i], self.xw[i], self.rw[i], W[i].nx, W[i].h, self.N1, self.coarse_ordering)
        # reference: range(2, nw) -> the coarsest correction climbs to xw[1] but never into xw[0]
        for i in range(2, nw + (1 if self.fix_coarse_ascent else 0)):
            c, f = nw - i + 1, nw - i
            o.prolong_add(self.xw[c], W[c].nx, self.xw[f], W[f].nx, 0, W[f].ny, 0, 0)
            o.sor(self.bw[f], self.xw[f], self.rw[f], W[f].nx, W[f].h, self.N2, self.coarse_ordering)

    def _coarse_solve(self):
        """Gather the staging-level right-hand side to rank 0, run the tail, scatter the correction
        into the staging slab (which then plays the role of the reference's split(x_wl[0]), :307)."""
        o, P = self.ops, self.plan
        S = P.levels[self.n_para]
        if self.world == 1:
            self._run_tail()
            return self._xw0_slab
        stage = self.b[self.n_para]
        if self.rank == 0:
            parts = list(self._gather_buf.view(self.world, S.ny, S.ld).unbind(0))
            self.comm.gather0(stage, parts)
            R = S.rows
            with o.ctx():
                g = self._gather_buf.view(self.world, S.ny, S.ld)
                # owned rows of every rank, in order; the last global row is boundary (stays zero)
                self.bw[0][: self.world * R].view(self.world, R, S.ld).copy_(g[:, 1 : R + 1, :])
            self._run_tail()
            with o.ctx():
                # rank p gets global rows p*R-1 .. (p+1)*R of the correction = padded rows p*R .. p*R+R+1
                win = self._xw0_pad.as_strided((self.world, S.ny, S.ld), (R * S.ld, S.ld, 1))
                self._scatter_buf.view(self.world, S.ny, S.ld).copy_(win)
            parts = list(self._scatter_buf.view(self.world, S.ny, S.ld).unbind(0))
            self.comm.scatter0(stage, parts)
        else:
            self.comm.gather0(stage, None)
            self.comm.scatter0(stage, None)
        return stage

    def _run_tail(self):
        if not self.use_graph or self.world == 1:  # on one rank the tail is part of the cycle graph
            self._tail()
            return
        if self._tail_graph is None:
            self.ops.graph_begin()
            self._tail()
            self._tail_graph = self.ops.graph_end()
        self.ops.graph_launch(self._tail_graph)

    def _cycle(self):
        """One V-cycle (the body of the loop at parallel_multigrid.py:270-347), norm included."""
        o, P, n_para = self.ops, self.plan, self.n_para
        Ls = P.levels
        # ---- descent
        self._exchange(self.r[0], Ls[0], up=True, down=False)
        o.restrict(self.r[0], Ls[0].nx, self.b[1], Ls[1].nx, Ls[1].t0, Ls[1].t1, 0, -1)
        for l in range(1, n_para):
            L = Ls[l]
            xa, xb = self._xa[l], self._xb[l]
            if self.fused:  # x = 0, N1 sweeps and the residual in one pass (result must land in xa)
                res = self._smooth_fused(l, None, None, 0, self.N1, self.r[l])
                if res is not xa:
                    o.copy_(xa, res)
                o.restrict(self.r[l], L.nx, self.b[l + 1], Ls[l + 1].nx, Ls[l + 1].t0, Ls[l + 1].t1, 0, -1)
                continue
            # x = 0 then N1 sweeps: the first sweep needs no x (all zeros), the rest ping-pong
            o.jacobi_from_zero(self.b[l], xa, L.nx, L.t0, L.t1, L.h)
            self._exchange(xa, L)
            res = self._jacobi_sweeps(l, xa, xb, self.N1 - 1)
            if res is not xa:  # only when N1 - 1 is odd: bring the result back into xa
                o.copy_(xa, res)
            o.residual(self.r[l], xa, self.b[l], L.nx, L.t0, L.t1, L.h)
            self._exchange(self.r[l], L, up=True, down=False)
            o.restrict(self.r[l], L.nx, self.b[l + 1], Ls[l + 1].nx, Ls[l + 1].t0, Ls[l + 1].t1, 0, -1)
        # ---- coarse
        corr = self._coarse_solve()
        # ---- ascent
        for l in range(n_para - 1, 0, -1):
            L = Ls[l]
            xa, xb = self._xa[l], self._xb[l]
            cnx = Ls[l + 1].nx
            if self.fused:
                # the descent left x_l in xa; the result may land in either array, the next level up
                # reads it from wherever it is and the next descent starts from zero anyway
                corr = self._smooth_fused(l, xa, corr, cnx, self.N2, self.r[l] if self.keep_ascent_residuals else None)
                continue
            o.prolong_add(corr, cnx, xa, L.nx, L.t0, L.t1, 0, -1)
            self._exchange(xa, L)
            res = self._jacobi_sweeps(l, xa, xb, self.N2)
            assert res is xa
            if self.keep_ascent_residuals:
                o.residual(self.r[l], xa, self.b[l], L.nx, L.t0, L.t1, L.h)
            corr = xa
        self._smooth_level0(self.N2, want_norm=True, crs=corr)

    # ------------------------------------------------------------------ drivers
    def _run_cycle(self):
        if self.use_graph and self.world == 1:
            # one graph per array the finest x starts the cycle in; a cycle may leave it in the other
            key = self._cur0
            if key not in self._graphs:
                self.ops.graph_begin()
                self._cycle()
                self._graphs[key] = (self.ops.graph_end(), self._cur0)
                self._cur0 = key
            g, end = self._graphs[key]
            self.ops.graph_launch(g)
            self._cur0 = end
        else:
            self._cycle()

    def _global_err(self):
        if self.world == 1:
            return self.ops.norm_read(self.norm)
        with self.ops.ctx():
            v = self.ops.norm_tensor(self.norm)
            # a NaN on any rank must win, whatever the backend's max does with NaNs
            t = torch.cat([torch.nan_to_num(v, nan=0.0), torch.isnan(v).to(v.dtype)])
        t = self.comm.allreduce_max(t)
        with self.ops.ctx():
            vals = t.tolist()
        return float("nan") if vals[1] > 0 else float(vals[0])

    def cycles(self, k):
        """Run exactly k V-cycles (no stop test); returns max|r| after each."""
        errs = []
        self.ops.begin()
        for _ in range(k):
            self._run_cycle()
            errs.append(self._global_err())
        self.ops.end()
        return errs

    def presmooth(self):
        """The single Jacobi sweep + residual that opens solve() (parallel_multigrid.py:264-266)."""
        self.ops.begin()
        self._smooth_level0(1, want_norm=False)
        self.ops.end()

    def solve(self):
        """Multigrid.solve(): returns max|r| over the whole grid after the last cycle."""
        self.it = 0
        self.err_history = []
        self.diverged = False
        err, err_old = 0.0, 100.0
        self.presmooth()
        self.ops.begin()
        done = False
        while not done and self.it < self.max_cycles:
            self._run_cycle()
            err = self._global_err()
            self.err_history.append(err)
            if err > err_old * 1.1:
                print("Convergence compromised : r>r+1, set n1 higher")
                self.diverged = True
            err_old = err
            done = not (err > self.epsilon and not self.diverged)
            self.it += 1
        self.ops.end()
        return err

    def close(self):
        for g, _ in self._graphs.values():
            self.ops.graph_destroy(g)
        self._graphs = {}
        if self._tail_graph is not None:
            self.ops.graph_destroy(self._tail_graph)
            self._tail_graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
