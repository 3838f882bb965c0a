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
* the four overlapping 2x2 blocks become P row slabs of the global grid (one per GPU/rank) with H
  halo rows on each side; what crosses a slab boundary is a contiguous block of rows per neighbour;
* restriction/prolongation need no per-rank offsets table: with slab boundaries on even rows the
  fine centre of local coarse row T is local fine row 2T-H on every rank (ofst_j = -H, ofst_i = 0);
* a level's smoothing (x = 0 or the prolongation, its 3 or 6 Jacobi sweeps, the residual) is ONE
  fused pass over memory (mg_smooth_fused); with several ranks the halo is as deep as the pass
  (H = 8), each rank recomputes the few rows next to its slab that the pass needs, and the halo
  is exchanged once per pass instead of once per sweep;
* only the Jacobi levels whose slabs are tall enough to be worth an exchange are distributed; from
  the first level with fewer than `dist_min_rows` rows per rank on, one all-gather of the right-hand
  side gives every rank the whole grid, and every rank runs the remaining Jacobi levels and the
  whole-domain levels redundantly, without any communication, as one hipGraph (the reference also
  replicates its whole-domain levels on every rank, behind four broadcasts, :117-138).  The
  correction a rank needs is then a row window of its own copy;
* Jacobi sweeps ping-pong between two arrays instead of copying a scratch array back;
* on one GPU the whole V-cycle is captured once into a hipGraph and replayed per cycle.

The engine is written against an ``ops`` object (multigrid_amd.ops.HipOps: the HIP kernels behind
the C ABI) and a torch.distributed process group for the exchanges.
"""
import time
import warnings

import torch
import torch.distributed as dist

from .comm import SlabComm
from .levels import choose_n_dist, make_plan, whole_level_fits_lds


class SlabMultigrid:
    N1 = 3  # pre-smoothing sweeps  (parallel_multigrid.py:255)
    N2 = 6  # post-smoothing sweeps (parallel_multigrid.py:256)

    def __init__(self, gny, gnx, n_para, h0, epsilon, *, ops=None, group=None, world=None, rank=None,
                 coarse_ordering="lexicographic", fix_coarse_ascent=False, keep_ascent_residuals=False,
                 max_cycles=100, use_graph=True, fused=True, halo=None, eager_dead_levels=False,
                 store_residual_every_cycle=False, batch=None, dist_min_rows=None, recompute_presmoothing=True,
                 recompute_min_nx=1024, recompute_min_blocks=1024, jacobi_omega=1.0, tune_placement=None,
                 overlap_min_rows=4096, batch_stop="max", fine_smoother="jacobi", comm=None,
                 recompute_last_distributed=True, dist_levels=None, t_round=None, gather_bw=None, graph_comm=False):
        if ops is None:
            from .ops import HipOps
            ops = HipOps()
        self.ops = ops
        self.group = group
        if world is None:
            world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
            rank = dist.get_rank(group) if world > 1 else 0
        self.world, self.rank = world, rank
        # fine_smoother: "jacobi" = the reference's sweep on the Jacobi levels; "redblack" = red-black Gauss-Seidel
        # sweeps in their place (same counts n1/n2; damps the checkerboard mode that undamped Jacobi leaves alone).
        # One kernel per colour and operation (the fused passes are Jacobi pipelines); on several ranks one halo row is
        # exchanged after every colour.
        if fine_smoother not in ("jacobi", "redblack"):
            raise ValueError("fine_smoother must be 'jacobi' or 'redblack'")
        if fine_smoother == "redblack":
            fused = False
        self.fine_smoother = fine_smoother
        self.fused = fused and getattr(ops, "supports_fused", False)
        if halo is None and world > 1:
            # fused: as deep as the deepest pass (6 sweeps + residual + restriction = 8; the 9-sweep pass of
            # the recompute scheme: 10, rounded to 12 so that the halo of every level stays even);
            # one exchange per sweep needs one row, 2 keeps the halo even
            halo = (12 if recompute_presmoothing else 8) if self.fused else 2
        # Jacobi levels 0 .. n_dist-1 are distributed, the others replicated on every rank: as many as the
        # cost model of levels.choose_n_dist finds cheapest (exchange rounds against the all-gather of the
        # first replicated level and the redundant work on the replicated ones), or, with dist_min_rows,
        # every level with at least that many rows per rank.
        n_dist = n_para
        if world > 1:
            R0 = (gny - 1) // world
            if dist_levels is not None:  # the caller's choice; make_plan refuses what cannot be cut
                n_dist = max(1, min(n_para, int(dist_levels)))
            elif dist_min_rows is None:
                # t_round, gather_bw: measured cost of one exchange round (s) and rate of the all-gather (B/s) on this
                # machine, where the caller has them (bench.py measures both); None = the cost model's defaults
                n_dist = choose_n_dist(gny, gnx, n_para, world, halo, t_round=t_round, gather_bw=gather_bw)
            else:
                n_dist = max(1, sum(1 for l in range(n_para) if (R0 >> l) >= max(dist_min_rows, halo)))
        self.n_dist = n_dist
        self.plan = make_plan(gny, gnx, n_para, world, rank, h0, halo, n_dist=n_dist)
        # the same grid on one rank, with the same halo depth: layout of the replicated levels; a
        # rank's slab window of such an array (its rows plus the halo rows) is a row range of it
        self.gplan = self.plan if world == 1 else make_plan(gny, gnx, n_para, 1, 0, h0, self.plan.halo)
        self.lv = [(self.plan if l < n_dist else self.gplan).levels[l] for l in range(n_para + 1)]
        W0 = self.plan.whole[0]
        if rank == 0 and coarse_ordering == "lexicographic" and not whole_level_fits_lds(W0.ny, W0.nx):
            warnings.warn(f"the first whole-domain level ({W0.ny} x {W0.nx}) does not fit LDS: its SSOR wavefront will walk "
                          "global memory, an order of magnitude slower per step; use more Jacobi levels "
                          "(levels.choose_n_para) or coarse_ordering='redblack'")
        # comm: an object with SlabComm's interface in its place (tools/rank_rehearsal.py runs one rank of an
        # N-rank job alone on one GPU that way, with the rank as its own neighbours)
        self.comm = comm if comm is not None else SlabComm(ops, group, world, rank)
        self.n_para = n_para
        self.epsilon = epsilon
        self.coarse_ordering = coarse_ordering
        self.fix_coarse_ascent = fix_coarse_ascent
        self.keep_ascent_residuals = keep_ascent_residuals
        self.max_cycles = max_cycles
        # 1 = the reference's Jacobi sweep.  It leaves the checkerboard mode of the error undamped (eigenvalue
        # -1) and the restriction does not see that mode, so max|r| stalls at the mode's share of the
        # right-hand side; 0 < jacobi_omega < 1 (0.8 is the usual choice) damps every sweep,
        # x + omega * (t - x), and lets the cycle converge to rounding level.
        if not 0 < jacobi_omega <= 1:
            raise ValueError("jacobi_omega must be in (0, 1]")
        self.jacobi_omega = float(jacobi_omega)
        # On a level that is not cut into slabs the post-smoothing pass recomputes the pre-smoothing sweeps
        # (from x = 0, 3 sweeps) in front of its own instead of reading their result: the descent pass then
        # need not store x at all.  16 B per point less traffic on every coarse level, identical values.
        self.recompute_presmoothing = recompute_presmoothing
        self.recompute_min_nx, self.recompute_min_blocks = recompute_min_nx, recompute_min_blocks
        self.recompute_last_distributed = recompute_last_distributed
        # With the reference's loop bounds nothing computed on the whole-domain levels below the first
        # ever reaches the solution (:302).  They are a pure function of rw[0], so unless asked to they
        # are not computed every cycle but on demand (sync_whole_levels), with identical values.
        self.eager_dead_levels = eager_dead_levels or fix_coarse_ascent
        self.use_graph = use_graph and getattr(ops, "supports_graphs", False)
        # One graph per cycle: always on one rank.  On several ranks the distributed levels are launched one by one
        # with the exchanges between them, unless graph_comm is set and the exchanges are the library's raw RCCL calls
        # (comm.mode == "native"), which hipGraph capture records like kernels: the whole cycle, exchanges and the
        # all-gather included, is then one graph launch per rank (the MAX-reduction of the norm stays outside), with
        # every exchange on the cycle's main stream: passes are not cut around their exchanges in this mode.
        # Every rank must make the same choice.
        self.graph_comm = bool(graph_comm) and world > 1 and self.use_graph and getattr(self.comm, "mode", None) == "native"
        self._whole_graph = self.use_graph and (world == 1 or self.graph_comm)
        self.it = 0
        self.err_history = []
        self.diverged = False

        P = self.plan
        H = P.halo
        # batch = B: B independent (gny x gnx) problems advance together, every level array gets a
        # leading batch dimension and every launch covers all of them (one rank only).  The norm and
        # the stop rule are taken over the whole batch.
        if batch is not None and world != 1:
            raise ValueError("batched solves run on one rank")
        self.batch = batch
        # batch_stop: what solve() applies the reference's stop rule (:336-345) to.  "max": the maximum of max|r|
        # over the batch, all problems run the same number of cycles.  "each": every problem on its own -- it stops
        # at the cycle at which its own solve() would stop and keeps the x it had then (the batch goes on until the
        # last problem is done; a finished problem's further cycles are computed and discarded).
        if batch_stop not in ("max", "each"):
            raise ValueError("batch_stop must be 'max' or 'each'")
        if batch_stop == "each" and (batch is None or not self.fused):
            raise ValueError("batch_stop='each' is for batched solves with fused passes")
        self.batch_stop = batch_stop
        self.problem_its = self.problem_errs = self.problem_diverged = None
        z = (lambda rows, ld: ops.zeros(rows, ld, batch)) if batch is not None else ops.zeros
        # Jacobi levels: two x arrays (ping-pong), b, r.  b has one more level (the staging level).
        LV = self.lv
        # the finest level's arrays first: x (two, ping-pong), b, r
        self.placement = None  # what _place_finest measured (None: placement not tuned)
        fin = None
        self.placement_seconds = 0.0
        t0 = time.perf_counter()
        # tune_placement: None = wherever it matters and is safe -- finest arrays of at least 1 GiB (_place_finest returns
        # at once below that) on a GPU this rank has to itself (ranks that share one GPU over gloo, as the rehearsal
        # tests do, would take each other's memory with their candidate arrays); True / "verbose" / False = the caller's word
        if tune_placement is None:
            tune_placement = world == 1 or not getattr(self.comm, "staged", False)
        if tune_placement and batch is None and getattr(ops, "supports_placement", False) and self.fused:
            fin = self._place_finest(verbose=tune_placement == "verbose")
        cb1 = xa1 = None  # the array the finest pass restricts into and level 1's x, if the placement selection made them
        if fin is None:
            fin = [z(LV[0].ny, LV[0].ld) for _ in range(4)]
        else:
            fin, cb1, xa1 = fin
        slab1 = world > 1 and n_dist == 1  # the finest pass restricts into the slab that is then all-gathered
        self._xa = [fin[0]] + [xa1 if (l == 1 and xa1 is not None) else z(L.ny, L.ld) for l, L in enumerate(LV[:n_para]) if l >= 1]
        self._xb = [fin[1]] + [z(L.ny, L.ld) for L in LV[1:n_para]]
        self.b = [fin[2]] + [cb1 if (l == 1 and cb1 is not None and not slab1) else z(L.ny, L.ld) for l, L in enumerate(LV) if l >= 1]
        self.r = [fin[3]] + [z(L.ny, L.ld) for L in LV[1:n_para]]
        # the last distributed level restricts into a slab, which is all-gathered into b[n_dist]
        self._bslab = (cb1 if (slab1 and cb1 is not None) else z(P.levels[n_dist].ny, P.levels[n_dist].ld)) if world > 1 else None
        self._cur0 = 0  # which of (_xa[0], _xb[0]) holds the finest x
        # two accumulators, used in turn (the one the finest x's parity selects): the host can read a cycle's
        # norm while the next cycle, which resets and fills the other one, is already running
        if batch_stop == "each":
            self._norms = [ops.norm_slots_per_problem(batch), ops.norm_slots_per_problem(batch)]
        else:
            self._norms = [ops.norm_slots(), ops.norm_slots()]
        self.norm = self._norms[0]
        # Whole-domain levels, on every rank.  The first one shares the rows of the (replicated)
        # staging level: local row t of the padded arrays is global row t - H.
        self.xw, self.bw, self.rw = [], [], []
        S, W0 = LV[n_para], P.whole[0]
        self._xw0_pad = z(S.ny, S.ld)
        self.xw.append(self._xw0_pad[..., H : H + W0.ny, :])
        self.bw.append(self.b[n_para][..., H : H + W0.ny, :])
        # the same window of an equally padded array: one batch stride for all arrays of a launch
        self.rw.append(z(S.ny, S.ld)[..., H : H + W0.ny, :])
        for W in P.whole[1:]:
            self.xw.append(z(W.ny, W.ld))
            self.bw.append(z(W.ny, W.ld))
            self.rw.append(z(W.ny, W.ld))
        # Several ranks, fused passes: a pass that follows a halo exchange is cut into the rows that depend
        # on owned rows only -- launched first; the exchange runs beside them on a stream of its own --
        # and the few rows at both slab ends, launched (as one launch, on that stream) once the halo rows have
        # arrived.  Done on levels with at least overlap_min_rows rows per rank (None: never); on smaller ones
        # the interior rows are too short-lived to hide an exchange behind (measured: 4096 and 2048 run alike,
        # 1024 and below slower).
        # (not with graph_comm: RCCL calls are recorded correctly on the capturing stream itself, but a send/receive group
        # issued on a stream that joined the capture through an event -- the communication stream of a cut pass -- ends
        # the process inside the capture on RCCL 2.26 / HIP 7.0, tools/rccl_graph_probe2.py)
        self.overlap_min_rows = overlap_min_rows if (world > 1 and self.fused and hasattr(ops, "comm_side") and not self.graph_comm) else None
        self.split_passes = 0  # passes cut this way so far (diagnostics)
        self._pending = None  # (level, exchange items) of a right-hand side whose halo is still to be filled
        self._graphs = {}
        self._xchg_cache = {}
        self._mid_graph = None
        self._dead_graph = None
        self._b0_halo_valid = False
        self._r0_stale = False
        # store the finest residual in every cycle's pass instead of once at the end of solve()/cycles()
        self.store_residual_every_cycle = store_residual_every_cycle
        if tune_placement:
            self.placement_seconds = time.perf_counter() - t0

    PLACE_MAX_CANDIDATES = 26
    PLACE_SPREAD_CANDIDATES = 14  # further candidates allocated (not timed one by one) to reach another stretch of the sequence
    PLACE_BUDGET_SECONDS = 1.5    # no new candidate is allocated after this long (allocating 8 GiB takes 5 to 200 ms)

    def _place_finest(self, verbose=False):
        """Choose the finest level's arrays among freshly allocated candidates by how fast the cycle's finest
        pass runs when it writes into them.

        The same launch of that pass takes 5.4 or 6.1 ms at 32769 x 32769 depending on which allocations it
        WRITES: x, and the coarse right-hand side, although that is a fourteenth of the bytes; the arrays it only
        reads matter by 2-3 %.  What decides is the combination of the two written allocations: with the coarse
        right-hand side fixed, in a fresh process the first eleven 8-GiB allocations are slow as x and all later
        ones fast; with x fixed among the fast ones, a coarse right-hand side allocated after them makes the pass
        slow again and one allocated before them fast.  It is fixed when the allocations are made and survives
        the release of others; distances between the arrays inside one allocation, tile shapes and the XCD
        mapping do not matter; pure reads and pure writes of every array run at the same rate; the address-
        translation and L2 counters of a slow and a fast launch agree (tools/placement_*.py, whose outputs are kept
        under profiles/).  The cause below the allocation interface was not found, so set-up measures:
        the coarse right-hand side is allocated first AND KEPT (it becomes the engine's own), candidates of the
        finest array's size are allocated one by one, each is timed as the output of the pass, and as soon as
        both kinds have been seen and four candidates are of the fast kind those become x (two arrays), b and r
        (in the assignment whose pass, timed in both directions of x, is fastest).  About half a second.  Values do not depend on any of it.
        Returns (the four arrays, the coarse right-hand side) or None when the level is small (< 1 GiB per
        array) or memory is short."""
        eng, ops, L, Lc = self, self.ops, self.lv[0], self.plan.levels[1]
        nbytes = L.ny * L.ld * 8
        if nbytes < (1 << 30):
            return None
        free = torch.cuda.mem_get_info(ops.device)[0]
        # room for the other levels (a third of the finest's four arrays, rounded up to two arrays) and a margin
        kmax = min(self.PLACE_MAX_CANDIDATES, (free - 2 * nbytes - (8 << 30)) // nbytes)
        if kmax < 6:
            return None
        cb = ops.zeros(Lc.ny, Lc.ld)  # what the finest pass restricts into: kept
        crs, norm = ops.zeros(Lc.ny, self.lv[1].ld), ops.norm_slots()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

        t_begin = time.perf_counter()

        def time_pass(b, x_in, x_out, crs=crs):
            ops.begin()
            for rep in range(3):  # the first launch is not timed
                ops.smooth_fused(b, x_in, x_out, None, crs, L.nx, eng.lv[1].nx, L.t0, L.t1, L.fix_lo, L.fix_hi,
                                 L.oj, eng.N2, L.h, norm, cb, Lc.nx, Lc.t0, Lc.t1, omega=eng.jacobi_omega)
                if rep == 0:
                    ev[0].record(ops.stream)
            ev[1].record(ops.stream)
            ops.stream.synchronize()
            return ev[0].elapsed_time(ev[1]) / 2

        arrs, ms, fast, gap = [], [], [], False
        try:
            while len(arrs) < kmax:
                arrs.append(ops.zeros(L.ny, L.ld))
                if len(arrs) < 3:
                    continue
                for k in range(len(ms), len(arrs)):
                    i0, i1 = [j for j in range(3) if j != k][:2] if k < 2 else (0, 1)
                    ms.append(time_pass(arrs[i0], arrs[i1], arrs[k]))
                    if verbose:
                        print(f"placement: candidate {k} at GiB {arrs[k].data_ptr() / 2 ** 30:.3f}: {ms[k]:.3f} ms as the "
                              "output of the finest pass", flush=True)
                lo = min(ms)
                fast = [k for k in range(len(ms)) if ms[k] <= 1.03 * lo]
                gap = max(ms) > 1.06 * lo  # both kinds have been seen (they are 10 % or more apart)
                if gap and len(fast) >= 4:
                    break
                if len(ms) >= 4 and time.perf_counter() - t_begin > self.PLACE_BUDGET_SECONDS:
                    break
        except torch.OutOfMemoryError:  # someone else took the memory meanwhile: choose among what there is
            torch.cuda.empty_cache()
        if len(ms) < 4:
            del arrs
            torch.cuda.empty_cache()
            return None
        pick = sorted(range(len(ms)), key=lambda k: ms[k])[:4]  # x (two arrays), b, r
        # the cycle's own combination: x moves between two of them, b is a third.  Where the kinds are mixed, two
        # candidates that are fast as outputs beside the first two as inputs may still make a slow pair: try the
        # other assignments of the four then.
        combos, best = 0, None
        for ia, ib, ic in ((0, 1, 2), (0, 2, 1), (1, 2, 0), (0, 3, 1), (1, 3, 0), (2, 3, 0)):
            xa, xb, bb = arrs[pick[ia]], arrs[pick[ib]], arrs[pick[ic]]
            t = 0.5 * (time_pass(bb, xa, xb) + time_pass(bb, xb, xa))
            combos += 1
            if best is None or t < best[0]:
                best = (t, ia, ib, ic)
            if t <= 1.02 * min(ms):
                break
        final, ia, ib, ic = best
        pick = [pick[ia], pick[ib], pick[ic], pick[6 - ia - ib - ic]]
        # Spread over stretches of the allocation sequence.  Measured with thirty candidates: the pass is fastest
        # (by 2-4 % on top of the above) when the array it writes shares its stretch of ~96 GiB with neither of the
        # two big arrays it reads -- b from the early stretch (the kind that is slow as an output reads as fast as any),
        # the two x arrays from two different later stretches.  So: b := a candidate of the slow kind; one x := the
        # fastest output; the other x := whichever of some candidates further along the sequence makes the fastest
        # pair, timed in both directions as the cycle uses them.  Kept only if faster than the choice above.
        spread = None
        slow = [k for k in range(len(ms)) if ms[k] > 1.06 * min(ms)]
        if gap and slow and len(arrs) < kmax:
            n_timed = len(arrs)
            try:
                while (len(arrs) < min(kmax, n_timed + self.PLACE_SPREAD_CANDIDATES)
                       and time.perf_counter() - t_begin < self.PLACE_BUDGET_SECONDS):
                    arrs.append(ops.zeros(L.ny, L.ld))
            except torch.OutOfMemoryError:
                torch.cuda.empty_cache()
            later = list(range(n_timed, len(arrs)))
            xo, bk, partners = self._spread_plan(pick, slow, later)
            for xi in partners:
                t = 0.5 * (time_pass(arrs[bk], arrs[xi], arrs[xo]) + time_pass(arrs[bk], arrs[xo], arrs[xi]))
                combos += 1
                if verbose:
                    print(f"placement: b = candidate {bk}, x = candidates {xo} and {xi}: {t:.3f} ms", flush=True)
                if spread is None or t < spread[0]:
                    spread = (t, xi)
            if spread is not None and spread[0] < 0.995 * final:
                rk = next((k for k in slow[1:] + list(range(len(arrs))) if k not in (xo, spread[1], bk)))
                final, pick = spread[0], [xo, spread[1], bk, rk]
        if len(set(pick)) != 4:  # never two roles in one array (x and b in one tensor would make every solve wrong)
            del arrs, xa, xb, bb
            torch.cuda.empty_cache()
            return None
        fin = [arrs[k] for k in pick]
        self.placement = {"candidates": len(ms), "ms_as_output": [round(t, 3) for t in ms], "chosen": pick,
                          "both_kinds_seen": gap, "ms_chosen_combination": round(final, 3), "combinations_timed": combos,
                          "spread_over_the_allocation_sequence": spread is not None and pick[1] == spread[1] and pick[2] in slow}
        # level 1's x is the correction the finest pass reads: its candidates are judged by both passes.  They are
        # allocated while the candidates above are still held, so that they come from further along the sequence and
        # not from the memory the unchosen ones give back (which is the stretch the level's right-hand side lies in).
        xa1 = self._place_level1(cb, ev, lambda c: 0.5 * (time_pass(fin[2], fin[0], fin[1], c) + time_pass(fin[2], fin[1], fin[0], c)),
                                 t_begin, verbose)
        del arrs, xa, xb, bb
        torch.cuda.empty_cache()
        for a in fin + [cb]:  # the timed launches left values behind: back to zero
            ops.zero_(a)
        ops.stream.synchronize()
        return fin, cb, xa1

    @staticmethod
    def _spread_plan(pick, slow, later):
        """The spread step's candidates: (xo, bk, partners) -- xo the x array that stays (the fastest output), bk the
        candidate of the slow kind that becomes b, partners the candidates tried as the other x (the second of the
        four chosen so far, then some of the `later` allocations), none of them xo or bk and none twice.  partners is
        empty when xo and bk are one candidate (only one fast candidate was seen: nothing to spread)."""
        xo, bk = pick[0], slow[0]
        partners = []
        if xo != bk:
            for xi in [pick[1]] + later[len(later) // 2 :: 3] + later[-1:]:
                if xi not in partners and xi not in (xo, bk):
                    partners.append(xi)
        return xo, bk, partners

    def _place_level1(self, b1, ev, finest_with, t_begin, verbose=False):
        """The same for the one other pass that writes gigabytes: level 1's ascent pass (its x, a quarter of the
        finest array) takes 1.02 or 1.10 ms at 32769 x 32769 depending on the allocation it writes.  Up to eight
        candidates are timed as its output (b = the level's right-hand side, which is kept) and as the correction the
        finest pass reads (finest_with(candidate): ms of that pass on the arrays chosen for it); the one with the
        smallest sum becomes the level's x.  Only where that pass is the 9-stage one (recomputed pre-smoothing); None
        otherwise."""
        if self.n_para < 3 or not self._recomputed(1) or (self.world > 1 and self.n_dist < 2):
            return None
        eng, ops, L1, L2 = self, self.ops, self.lv[1], self.lv[2]
        crs = ops.zeros(L2.ny, L2.ld)
        cands, ms, fin_ms = [], [], []
        try:
            for _ in range(8):
                a = ops.zeros(L1.ny, L1.ld)
                ops.begin()
                for rep in range(3):
                    ops.smooth_fused(b1, None, a, None, crs, L1.nx, L2.nx, L1.t0, L1.t1, L1.fix_lo, L1.fix_hi, L1.oj,
                                     eng.N1 + eng.N2, L1.h, None, prolong_after=eng.N1, omega=eng.jacobi_omega)
                    if rep == 0:
                        ev[0].record(ops.stream)
                ev[1].record(ops.stream)
                ops.stream.synchronize()
                cands.append(a)
                ms.append(ev[0].elapsed_time(ev[1]) / 2)
                fin_ms.append(finest_with(a))
                if len(ms) >= 6 and max(ms) > 1.05 * min(ms) and sum(t <= 1.01 * min(ms) for t in ms) >= 2:
                    break  # both kinds seen (they are 6-8 % apart), and the fast one twice
                if time.perf_counter() - t_begin > self.PLACE_BUDGET_SECONDS + 0.5:
                    break
        except torch.OutOfMemoryError:
            torch.cuda.empty_cache()
        if not fin_ms:
            return None
        k = min(range(len(fin_ms)), key=lambda q: ms[q] + fin_ms[q])
        best = cands[k]
        ops.zero_(best)
        ops.stream.synchronize()
        self.placement["level1_ms_as_output"] = [round(t, 4) for t in ms]
        self.placement["finest_ms_with_it_as_correction"] = [round(t, 3) for t in fin_ms]
        self.placement["level1_chosen"] = k
        del cands, crs
        torch.cuda.empty_cache()
        return best

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
        """Finest-level residual slab.  Fused passes compute the residual every cycle (for the norm and
        the restriction) but do not store it; it is written when a solve()/cycles() call ends or on
        demand (materialize_residual), with the values the per-cycle store would have left."""
        if self._r0_stale:
            if self.world > 1:
                raise RuntimeError("the residual array is not current: call materialize_residual() on every rank")
            self.materialize_residual()
        return self.r[0]

    def materialize_residual(self):
        """Write r = b - A x of the finest level for the current x (collective: one halo row of x is
        exchanged).  A no-op when the last pass stored it."""
        if not self._r0_stale:
            return
        L, x = self.plan.levels[0], self.x0
        self.ops.begin()
        self._exchange([(x, L, 1)])
        self.ops.residual(self.r[0], x, self.b[0], L.nx, L.t0, L.t1, L.h, None, True)
        self.ops.end()
        self._r0_stale = False

    def problem_errors(self):
        """max|r| of every problem of a batch after the last cycle (the stop rule itself looks at the
        maximum over the batch): a (batch,) tensor on the device."""
        if self.batch is None:
            raise ValueError("problem_errors() is for batched solves")
        r = self.local_rows(self.r0)
        self.ops.begin()
        with self.ops.ctx():
            per = r.abs().amax(dim=(1, 2))
        self.ops.end()  # the caller's stream sees the result
        return per

    def owned_rows(self, level=0):
        """(first, last+1) local rows this rank owns, the global boundary rows included."""
        L = self.plan.levels[level]
        return L.own0, L.own1

    def global_row_of_local(self, t, level=0):
        L = self.plan.levels[level]
        return self.rank * L.rows - L.halo + t

    def set_x_rows(self, rows):
        """Set the owned rows of the finest x (both ping-pong arrays, so boundary values agree)."""
        a, e = self.owned_rows()
        for arr in (self._xa[0], self._xb[0]):
            arr[..., a:e, : self.plan.gnx] = rows

    def set_b_rows(self, rows):
        a, e = self.owned_rows()
        self.b[0][..., a:e, : self.plan.gnx] = rows
        self._b0_halo_valid = False

    def local_rows(self, arr):
        """The owned rows of a finest-level array as a (rows x gnx) view."""
        a, e = self.owned_rows()
        return arr[..., a:e, : self.plan.gnx]

    # ------------------------------------------------------------------ halo exchange
    def _exchange(self, items, up=True, down=True):
        """items: (array, its SlabLevel, depth) triples.  Fill `depth` halo rows of every array from
        the neighbouring ranks, all in one batch: my upper halo <- the last owned rows of the rank
        above, my lower halo <- the first owned rows of the rank below.  `up`/`down` select which of
        my halos are filled.  Whole rows (padding included) are moved, so each message is one
        contiguous block."""
        if self.world == 1:
            return
        key = (tuple((arr.data_ptr(), depth) for arr, _, depth in items), up, down)
        batch = self._xchg_cache.get(key)
        if batch is not None:  # the arrays are allocated once: the same few batches recur every cycle
            self.comm.run(batch)
            return
        sends, recvs = [], []
        for arr, L, depth in items:
            H, R = L.halo, L.rows
            assert depth <= H and depth <= R
            if self.rank > 0:
                if down:  # the rank above needs my first owned rows as its lower halo
                    sends.append((arr[H : H + depth].view(-1), self.rank - 1))
                if up:
                    recvs.append((arr[H - depth : H].view(-1), self.rank - 1))
            if self.rank < self.world - 1:
                if up:    # the rank below needs my last owned rows as its upper halo
                    sends.append((arr[H + R - depth : H + R].view(-1), self.rank + 1))
                if down:
                    recvs.append((arr[H + R : H + R + depth].view(-1), self.rank + 1))
        self._xchg_cache[key] = self.comm.prepare(sends, recvs)
        self.comm.run(self._xchg_cache[key])

    def _overlaps(self, l, D):
        """Is a pass of depth D on level l cut into interior and slab ends around its exchange?"""
        if self.overlap_min_rows is None or self.world == 1 or l >= self.n_dist:
            return False
        R = self.plan.levels[l].rows
        return R >= self.overlap_min_rows and R >= 4 * (D + 4)

    def _defer_exchange(self, l, items):
        """The halo of level l's right-hand side, just produced, is needed by the level's descent pass, which
        comes next: leave the exchange to that pass, which hides it behind its interior rows."""
        self._flush_pending()
        self._pending = (l, items)

    def _take_pending(self, l):
        """The deferred exchange for level l's descent pass ([] if there is none); one that was meant for
        another consumer is carried out now."""
        if self._pending is not None and self._pending[0] == l:
            items, self._pending = self._pending[1], None
            return items
        self._flush_pending()
        return []

    def _flush_pending(self):
        if self._pending is not None:
            items, self._pending = self._pending[1], None
            self._exchange(items)

    def _pass(self, l, D, items, launch):
        """One fused pass over level l that reads D rows beyond the rows it produces, after the halo exchange
        `items`.  launch(j0, j1) enqueues the pass for local rows j0 <= j < j1; launch(j0, j1, (j2, j3)) for those
        and the rows j2 <= j < j3 in one launch."""
        L = self.lv[l]
        if not items or not self._overlaps(l, D):
            if items:
                self._exchange(items)
            launch(L.t0, L.t1)
            return
        # Interior: rows whose inputs, D rows up and down and the coarse rows around those, are owned rows.
        # The cuts are even rows (relative to the even halo depth): fine centres of coarse rows, so the
        # coarse rows of a fused restriction split with them.
        self.split_passes += 1
        a, e = self._cut(L, D)
        ready = self.ops.mark()         # the rows to be sent are final here
        launch(a, e)                    # enqueued first: the GPU has work while the host sets the exchange up
        with self.ops.comm_side(ready):
            self._exchange(items)
            # the slab ends follow the exchange on ITS stream: they run beside the interior rows (which write other
            # rows of the same arrays) instead of after them
            if getattr(self.ops, "supports_also_rows", False):
                launch(L.t0, a, (e, L.t1))  # both ends in one launch
            else:
                launch(L.t0, a)
                launch(e, L.t1)
        self.ops.comm_join()

    @staticmethod
    def _cut(L, D):
        """Local rows (a, e) between which a pass of depth D on slab level L depends on owned rows only: the
        stencil reaches D rows up and down, the prolongation one coarse row further; a and e are even
        relative to the (even) halo depth."""
        m = D + 2 + (D & 1)
        return L.halo + m, L.halo + L.rows - m

    @staticmethod
    def _coarse_rows(Lc, H, j0, j1):
        """Coarse rows whose fine centre row 2T - H is one of j0 .. j1-1, within the rows Lc updates."""
        return max(Lc.t0, (j0 + H + 1) // 2), min(Lc.t1, (j1 + H + 1) // 2)

    # ------------------------------------------------------------------ building blocks
    @staticmethod
    def _chunks(n, most=6):
        """n sweeps as passes of at most `most` sweeps, as even as possible (8 -> 4+4)."""
        k = (n + most - 1) // most
        return [n // k + (1 if i < n % k else 0) for i in range(k)]

    def _smooth(self, l, src, crs, n, r=None, norm=None, finest=False, crs_has_halo=False, restrict=False):
        """smooth_parallel (:100-114) on level l, optionally preceded by the prolongation of `crs`
        (:309, :322): [src += P(crs)], n Jacobi sweeps, [residual into r (+ norm)].  src None = start
        from x = 0 (:276).  The two x arrays of the level are used as ping-pong buffers; returns the
        one that holds the result.  Halo rows of the result are NOT valid.  restrict=True also
        produces b_{l+1} from the residual (:273, :288), fused into the pass when passes are fused (the
        residual is then stored only if `r` is given)."""
        L, o = self.lv[l], self.ops
        xa, xb = self._xa[l], self._xb[l]
        cnx = self.lv[l + 1].nx
        dist_ = self.world > 1 and l < self.n_dist           # this level is cut into slabs
        to_rep = dist_ and l + 1 == self.n_dist              # ... and the next one is replicated
        # where the restriction goes: the next level's b, or the slab that is then all-gathered into it
        Lc = self.plan.levels[l + 1] if to_rep else self.lv[l + 1]
        cb = self._bslab if to_rep else self.b[l + 1]
        H = L.halo

        def crows(j0, j1):  # the coarse rows a launch over fine rows j0 .. j1-1 restricts into
            return (Lc.t0, Lc.t1) if (j0, j1) == (L.t0, L.t1) else self._coarse_rows(Lc, H, j0, j1)

        def also_kw(also, restricts):  # a second row range of the same launch (the other slab end), with its coarse rows
            if also is None:
                return {}
            return {"also_rows": (*also, *(crows(*also) if restricts else (0, 0)))}

        def after_restriction():  # the next level's right-hand side has been produced
            if to_rep:
                self._allgather_b(l + 1)
            elif dist_:
                items = [(self.b[l + 1], Lc, self._b_halo_rows(l + 1))]
                # its first reader is the next level's descent pass: let that pass hide the exchange
                if l + 1 < self.n_dist and self._overlaps(l + 1, self.N1 + 2):
                    self._defer_exchange(l + 1, items)
                else:
                    self._exchange(items)

        pend = self._take_pending(l) if self.fused else []
        if self.fused and self._recomputed(l):
            if src is None:  # descent: only the restricted residual leaves the pass
                assert restrict and crs is None

                def launch(j0, j1, also=None):
                    o.smooth_fused(self.b[l], None, None, r, None, L.nx, cnx, j0, j1, L.fix_lo, L.fix_hi, L.oj, n, L.h,
                                   norm, cb, Lc.nx, *crows(j0, j1), omega=self.jacobi_omega, **also_kw(also, True))
                self._pass(l, n + 2, pend, launch)
                after_restriction()
                return None
            # ascent: N1 sweeps from zero, + P(crs), the n post-smoothing sweeps; of a slab's
            # neighbours only the correction's halo is needed (b's was exchanged when b was made)
            assert crs is not None and not restrict and not pend
            items = [(crs, self.lv[l + 1], (self.N1 + n) // 2 + 2)] if (dist_ and not crs_has_halo) else []

            def launch(j0, j1, also=None):
                o.smooth_fused(self.b[l], None, xa, r, crs, L.nx, cnx, j0, j1, L.fix_lo, L.fix_hi, L.oj, self.N1 + n, L.h,
                               norm, prolong_after=self.N1, omega=self.jacobi_omega, **also_kw(also, False))
            self._pass(l, self.N1 + n + (1 if r is not None else 0), items, launch)
            return xa
        if self.fused:
            extra = (2 if restrict else 1 if r is not None else 0)
            chunks = self._chunks(n, min(6, L.halo - extra) if dist_ else 6)
            for ci, k in enumerate(chunks):
                dst = xa if (src is None or src is xb) else xb
                first, last = ci == 0, ci == len(chunks) - 1
                depth = k + (extra if last else 0)
                items = pend if first else []
                if src is not None and dist_:  # the pass reads `depth` rows beyond the rows it produces
                    items = items + [(src, L, depth)]
                    if first and crs is not None and not crs_has_halo:
                        items.append((crs, self.lv[l + 1], depth // 2 + 2))
                if last and norm is not None:
                    o.norm_reset(norm)
                if last and restrict:
                    def launch(j0, j1, also=None, src=src, dst=dst, first=first, k=k):
                        o.smooth_fused(self.b[l], src, dst, r, crs if first else None, L.nx, cnx, j0, j1, L.fix_lo,
                                       L.fix_hi, L.oj, k, L.h, norm, cb, Lc.nx, *crows(j0, j1), omega=self.jacobi_omega,
                                       **also_kw(also, True))
                else:
                    def launch(j0, j1, also=None, src=src, dst=dst, first=first, last=last, k=k):
                        o.smooth_fused(self.b[l], src, dst, r if last else None, crs if first else None, L.nx, cnx, j0,
                                       j1, L.fix_lo, L.fix_hi, L.oj, k, L.h, norm if last else None, omega=self.jacobi_omega,
                                       **also_kw(also, False))
                self._pass(l, depth, items, launch)
                src = dst
            if restrict:
                after_restriction()
            return src
        # one kernel per operation, one single-row exchange per sweep (red-black: per colour, and in place)
        redblack = self.fine_smoother == "redblack"
        if src is None and redblack:
            src = xa
            o.zero_(src)
        elif src is None:
            o.jacobi_from_zero(self.b[l], xa, L.nx, L.t0, L.t1, L.h, finest, omega=self.jacobi_omega)
            src, n = xa, n - 1
        else:
            if crs is not None:  # its lower halo row is valid: last exchanged after its last sweep
                o.prolong_add(crs, cnx, src, L.nx, L.t0, L.t1, 0, L.oj)
        if dist_:
            self._exchange([(src, L, 1)])
        for _ in range(n):
            if redblack:
                for colour in (0, 1):  # the second colour reads the first one's values across the slab boundary
                    o.gauss_seidel_rb(self.b[l], src, L.nx, L.t0, L.t1, L.h, row_offset=L.halo, colour=colour)
                    if dist_:
                        self._exchange([(src, L, 1)])
                continue
            dst = xb if src is xa else xa
            o.jacobi(self.b[l], src, dst, L.nx, L.t0, L.t1, L.h, finest, omega=self.jacobi_omega)
            if dist_:
                self._exchange([(dst, L, 1)])
            src = dst
        if r is not None:
            if norm is not None:
                o.norm_reset(norm)
            o.residual(r, src, self.b[l], L.nx, L.t0, L.t1, L.h, norm, finest)
        if restrict:
            if dist_:  # full weighting needs one halo row of r above
                self._exchange([(self.r[l], L, 1)], up=True, down=False)
            o.restrict(self.r[l], L.nx, cb, Lc.nx, Lc.t0, Lc.t1, 0, L.oj)
            if to_rep:
                self._allgather_b(l + 1)
        return src

    def _b_halo_rows(self, l):
        """Halo rows of level l's right-hand side that its passes read: the descent pass N1 sweeps + residual +
        restriction deep, the ascent pass N2 sweeps (+ residual) -- or, where the pre-smoothing is recomputed, N1 + N2."""
        extra = 1 if self.keep_ascent_residuals else 0
        if self._recomputed(l):
            return self.N1 + self.N2 + extra
        return max(self.N1 + 2, self.N2 + extra)

    def _recomputed(self, l):
        """Does level l run in the recompute scheme?"""
        if not (self.recompute_presmoothing and self.fused and l >= 1 and self.N1 == 3 and self.N2 == 6
                and getattr(self.ops, "supports_prolong_after", False)):
            return False
        if self.world > 1 and l < self.n_dist and self.plan.halo < self.N1 + self.N2 + 1:
            return False  # a slab needs a halo as deep as both smoothings together
        if self.world > 1 and l + 1 == self.n_dist and self.recompute_last_distributed:
            # The last distributed level, whatever its size: its correction is a window of the replicated level below,
            # halo included, so in this scheme nothing is exchanged in front of its ascent pass -- one round less on
            # the cycle's critical path (a level this small cannot hide a round behind its interior rows), which
            # outweighs what the deeper pipeline costs on a narrow level.
            return True
        # The 9-deep pass pays with wider overlaps (20 of 128 columns, 18 rows per tile end), which a
        # level must be wide enough and have workgroups enough to absorb (measured: it wins from
        # 8193 x 8193 on, loses at 4097 x 4097 and on batches of narrow grids).
        L = self.lv[l]
        blocks = -(-(-(-L.nx // 108)) // 4) * -(-L.rows // 64) * (self.batch or 1)
        return L.nx >= self.recompute_min_nx and blocks >= self.recompute_min_blocks

    def rounds_per_cycle(self):
        """Communication rounds (neighbour exchanges and the all-gather) one V-cycle issues on this rank,
        the max-reduction of the norm not counted."""
        if self.world == 1:
            return 0
        k = self.n_dist
        if not self.fused:
            # per sweep one exchange (red-black: two), one after the start (zero or the prolongation), one of r before
            # the restriction; 1 + 6 + 1 on the finest level, (1 + 3 + 1) + (1 + 6) on the others; the all-gather
            if self.fine_smoother == "redblack":
                return 14 + 21 * (k - 1) + 1
            return 8 + 11 * (k - 1) + 1
        total = 2  # finest level: (x, correction) before its pass, b_1 (or the all-gather) after it
        for l in range(1, k):
            # after the descent pass: b_{l+1} halo or the all-gather.  Before the ascent pass: x and the
            # correction -- with the recompute scheme only the correction, and none if that is replicated
            total += 1 + (0 if (self._recomputed(l) and l + 1 == k) else 1)
        return total

    def _allgather_b(self, lc):
        """Every rank's owned rows of the slab right-hand side of level lc -> the replicated b[lc]."""
        H, R = self.plan.halo, self.plan.levels[lc].rows
        self.comm.allgather_rows(self.b[lc][H : H + self.world * R], self._bslab[H : H + R])

    def _window(self, arr, lc):
        """This rank's slab of a replicated level-lc array: its rows and the halo rows around them."""
        if self.world == 1:
            return arr
        R = self.plan.levels[lc].rows
        return arr[self.rank * R : self.rank * R + self.plan.levels[lc].ny]

    def _smooth_level0(self, n, want_norm, crs=None, store_r=False):
        xs = [self._xa[0], self._xb[0]]
        self._ensure_b0_halo()
        # the residual of the finest level is restricted right away: b_1 is ready for the next descent
        store_r = store_r or self.store_residual_every_cycle or not self.fused
        res = self._smooth(0, xs[self._cur0], crs, n, self.r[0] if store_r else None, self._norms[self._cur0] if want_norm else None,
                           finest=True, crs_has_halo=self.n_dist == 1, restrict=True)
        self._r0_stale = not store_r
        self._cur0 = 0 if res is xs[0] else 1

    def _tail(self):
        """First whole-domain level (parallel_multigrid.py:291-293); every rank runs it."""
        o, W = self.ops, self.plan.whole
        nw = len(W)
        o.zero_(self._xw0_pad)
        o.sor(self.bw[0], self.xw[0], self.rw[0], W[0].nx, W[0].h, self.N1, self.coarse_ordering)
        if self.fix_coarse_ascent:  # then the lower levels do feed xw[0]
            self._tail_below()

    def _has_dead_levels(self):
        return self.eager_dead_levels and not self.fix_coarse_ascent and len(self.plan.whole) >= 2

    def _dead_levels(self, after=None):
        """With the reference's loop bounds nothing in the cycle reads what the whole-domain levels
        below the first produce.  When they are computed every cycle they run beside the rest of
        the cycle, on the side stream; the cycle joins it at its end."""
        if not self._has_dead_levels():
            return
        o = self.ops
        with o.side(after):
            if self.use_graph and not self._whole_graph:  # else they are part of the cycle graph
                if self._dead_graph is None:
                    with o.capture() as cap:
                        self._tail_below()
                    self._dead_graph = cap.graph
                o.graph_launch(self._dead_graph)
            else:
                self._tail_below()

    def _tail_below(self):
        """The whole-domain levels below the first (parallel_multigrid.py:294-304)."""
        o, W = self.ops, self.plan.whole
        nw = len(W)
        for i in range(1, nw):
            o.zero_(self.xw[i])
            o.restrict(self.rw[i - 1], W[i - 1].nx, self.bw[i], W[i].nx, 1, W[i].ny - 1, 0, 0)
            o.sor(self.bw[i], self.xw[i], self.rw[i], W[i].nx, W[i].h, self.N1, self.coarse_ordering)
        # reference: range(2, nw) -> the coarsest correction climbs to xw[1] but never into xw[0]
        for i in range(2, nw + (1 if self.fix_coarse_ascent else 0)):
            c, f = nw - i + 1, nw - i
            o.prolong_add(self.xw[c], W[c].nx, self.xw[f], W[f].nx, 0, W[f].ny, 0, 0)
            o.sor(self.bw[f], self.xw[f], self.rw[f], W[f].nx, W[f].h, self.N2, self.coarse_ordering)

    def sync_whole_levels(self):
        """Bring xw[1:], bw[1:], rw[1:] to the state the reference's arrays have after the last cycle
        (no-op when they are computed every cycle)."""
        if not self.eager_dead_levels and len(self.xw) > 1:
            self.ops.begin()
            self._tail_below()
            self.ops.end()
            self.ops.synchronize()

    def _middle(self):
        """The part of the cycle that needs no communication: descent of the replicated Jacobi levels,
        the first whole-domain level, their ascent.  Every rank computes the same values.  Returns
        the array that holds the correction for the last distributed level."""
        n_para, k = self.n_para, self.n_dist
        for l in range(k, n_para):
            self._smooth(l, None, None, self.N1, None if self.fused else self.r[l], restrict=True)
        self._tail()
        corr = self._xw0_pad
        for l in range(n_para - 1, k - 1, -1):
            corr = self._smooth(l, self._xa[l], corr, self.N2, self.r[l] if self.keep_ascent_residuals else None,
                                crs_has_halo=True)
        return corr

    def _run_middle(self):
        # a from-zero pass always writes _xa[l] and an even number of passes follows on the ascent, so
        # the result lands in the same array every cycle: one captured graph serves all cycles
        if not self.use_graph or self._whole_graph:  # the middle is part of the cycle graph
            self._mid_result = self._middle()
        else:
            if self._mid_graph is None:
                with self.ops.capture() as cap:
                    self._mid_result = self._middle()
                self._mid_graph = cap.graph
            self.ops.graph_launch(self._mid_graph)
        # On one rank the side work is enqueued after the rest of the cycle (see _cycle), from this point of
        # the main sequence: captured in that order the graph runs the ascent first and the dead levels beside
        # it; captured right here, the ascent waited for most of them (0.26 ms per cycle at 32769 x 32769).
        if self.world == 1 or self._whole_graph:
            self._dead_mark = self.ops.mark() if self._has_dead_levels() else None
        else:
            self._dead_mark = None
            self._dead_levels()
        return self._mid_result

    def _cycle(self, store_r=False):
        """One V-cycle (the body of the loop at parallel_multigrid.py:270-347), norm included.
        store_r: the finest pass also writes the residual array (a caller that knows this is its
        last cycle saves the separate residual launch that way)."""
        k = self.n_dist
        # ---- descent of the distributed levels (b_1 was produced together with the finest residual, by
        # presmooth or the last cycle); the last one ends with the all-gather of b_k
        for l in range(1, k):
            self._smooth(l, None, None, self.N1, None if self.fused else self.r[l], restrict=True)
        # ---- everything below, on every rank alike
        corr = self._run_middle()
        # ---- ascent of the distributed levels: the first takes its window of the replicated result
        for l in range(k - 1, 0, -1):
            # x_l of the descent is in _xa[l] (a from-zero pass always writes there)
            rep = l + 1 == k
            corr = self._smooth(l, self._xa[l], self._window(corr, l + 1) if rep else corr, self.N2,
                                self.r[l] if self.keep_ascent_residuals else None, crs_has_halo=rep)
        if k == 1:
            corr = self._window(corr, 1)
        self._smooth_level0(self.N2, want_norm=True, crs=corr, store_r=store_r)
        if self._dead_mark is not None:
            self._dead_levels(after=self._dead_mark)
        self.ops.join_side()
        if self.graph_comm:
            # a captured cycle is the same every time it is replayed: the halo of the right-hand side the finest pass
            # has just restricted is exchanged here instead of being left to the next cycle's first pass
            self._flush_pending()

    # ------------------------------------------------------------------ drivers
    def _ensure_b0_halo(self):
        """The halo rows of the finest right-hand side, exchanged once after the caller has set it."""
        if not self._b0_halo_valid:  # b0 only changes when the caller sets it
            if self.fused and self.world > 1:
                self._exchange([(self.b[0], self.plan.levels[0], self.plan.halo - 1)])
            self._b0_halo_valid = True

    def _run_cycle(self, store_r=False):
        self._ensure_b0_halo()  # never inside a captured cycle, which is replayed for right-hand sides to come
        self._last_parity = self._cur0  # selects the norm accumulator this cycle fills
        store_r = bool(store_r) and self.fused and not self.store_residual_every_cycle
        if self._whole_graph:
            # one graph per array the finest x starts the cycle in (a cycle may leave it in the other)
            # and per residual-store variant
            key = (self._cur0, store_r)
            if key not in self._graphs:
                self._capture_graphs()
            g, end = self._graphs[key]
            self.ops.graph_launch(g)
            self._cur0 = end
            self._r0_stale = not (store_r or self.store_residual_every_cycle or not self.fused)
        else:
            self._cycle(store_r)

    def _capture_graphs(self):
        """Capture every variant of the cycle (capturing records, it does not execute): later cycles
        then never pay for a capture, whichever array the finest x is in."""
        cur, stale = self._cur0, self._r0_stale
        variants = [False] if (self.store_residual_every_cycle or not self.fused) else [False, True]
        for start in (0, 1):
            for store_r in variants:
                if (start, store_r) in self._graphs:
                    continue
                self._cur0 = start
                try:
                    with self.ops.capture() as cap:
                        self._cycle(store_r)
                except BaseException:
                    self._cur0, self._r0_stale = cur, stale
                    raise
                self._graphs[(start, store_r)] = (cap.graph, self._cur0)
        self._cur0, self._r0_stale = cur, stale

    def _reduce_err(self, parity=None):
        """Enqueue the MAX-reduction of a norm accumulator over the ranks (nothing to do on one).  The
        accumulator holds bit patterns of |r| as integers, an order that agrees with the order of the
        values and puts NaN above everything: one integer MAX-reduction of the slots gives every rank
        the global accumulator, NaN included."""
        if self.world > 1:
            self.comm.allreduce_max(self._norms[self._last_parity if parity is None else parity])

    def _read_err(self, parity=None, mark=None):
        """max|r| from a (reduced) accumulator: as it stands at `mark` of the main sequence, or now."""
        norm = self._norms[self._last_parity if parity is None else parity]
        if mark is not None:
            return self.ops.norm_read_at(norm, mark)
        return self.ops.norm_read(norm)

    def _global_err(self, parity=None, mark=None):
        """max|r| over all ranks."""
        self._reduce_err(parity)
        return self._read_err(parity, mark)

    def cycles(self, k):
        """Run exactly k V-cycles (no stop test); returns max|r| after each."""
        errs = []
        self.ops.begin()
        pending = None
        for c in range(k):
            self._run_cycle(store_r=c == k - 1)  # the last pass leaves the residual array current
            # Every cycle's norm is read, but after the next cycle has been enqueued, so the GPU does not
            # idle while the host waits for the number and issues the next cycle (on one rank a graph
            # launch; on several the launches and exchange rounds of the small distributed levels, which
            # the host issues more slowly than the GPU runs them: behind the previous cycle's finest pass
            # they cost nothing).  The reduction over the ranks is enqueued with its cycle.
            self._reduce_err()
            here = (self._last_parity, self.ops.mark())
            if pending is not None:
                errs.append(self._read_err(*pending))
            pending = here
            if self._cur0 == self._last_parity:  # the next cycle fills the same accumulator: read now
                errs.append(self._read_err(*pending))
                pending = None
        if pending is not None:
            errs.append(self._read_err(*pending))
        self._flush_pending()
        self.ops.end()
        self.materialize_residual()
        return errs

    def presmooth(self):
        """The single Jacobi sweep + residual that opens solve() (parallel_multigrid.py:264-266)."""
        self.ops.begin()
        self._smooth_level0(1, want_norm=False)
        self._flush_pending()
        self.ops.end()

    def _solve_each(self):
        """solve() with the stop rule applied to every problem of the batch on its own (batch_stop='each')."""
        import numpy as np
        B = self.batch
        self.it, self.err_history, self.diverged = 0, [], False
        its = np.zeros(B, dtype=np.int64)
        errs = np.zeros(B)
        err_old = np.full(B, 100.0)
        diverged = np.zeros(B, dtype=bool)
        active = np.ones(B, dtype=bool)
        kept = None  # x of the problems that are done, as it was when they stopped
        saved = np.zeros(B, dtype=bool)
        self.presmooth()
        self.ops.begin()
        while active.any() and self.it < self.max_cycles:
            self._run_cycle()
            per = self.ops.norm_read_per_problem(self._norms[self._last_parity])
            self.it += 1
            self.err_history.append(float("nan") if np.isnan(per).any() else float(per.max()))
            errs[active] = per[active]
            its[active] = self.it
            newly_div = active & (per > err_old * 1.1)
            if newly_div.any():
                print("Convergence compromised : r>r+1, set n1 higher")
            diverged |= newly_div
            err_old[active] = per[active]
            done_now = active & ~((per > self.epsilon) & ~diverged)  # NaN compares false: such a problem stops, as in solve()
            active &= ~done_now
            if done_now.any() and active.any():  # others go on: keep what the finished ones have now
                if kept is None:
                    kept = torch.empty_like(self.x0)
                idx = torch.from_numpy(np.flatnonzero(done_now)).to(self.x0.device)
                with self.ops.ctx():
                    kept[idx] = self.x0[idx]
                saved |= done_now
        if saved.any():
            idx = torch.from_numpy(np.flatnonzero(saved)).to(self.x0.device)
            with self.ops.ctx():
                for arr in (self._xa[0], self._xb[0]):  # both, so that boundary values keep agreeing
                    arr[idx] = kept[idx]
            self._r0_stale = True
        self._flush_pending()
        self.ops.end()
        self.materialize_residual()
        self.problem_its, self.problem_errs, self.problem_diverged = its, errs, diverged
        self.diverged = bool(diverged.any())
        return float(errs.max()) if not np.isnan(errs).any() else float("nan")

    def solve(self):
        """Multigrid.solve(): returns max|r| over the whole grid after the last cycle."""
        if self.batch_stop == "each":
            return self._solve_each()
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
        self._flush_pending()
        self.ops.end()
        self.materialize_residual()
        return err

    def close(self):
        self._xchg_cache = {}
        if getattr(self.comm, "close", None) is not None:
            self.comm.close()
        for g, _ in self._graphs.values():
            self.ops.graph_destroy(g)
        self._graphs = {}
        for name in ("_mid_graph", "_dead_graph"):
            if getattr(self, name, None) is not None:
                self.ops.graph_destroy(getattr(self, name))
                setattr(self, name, None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
