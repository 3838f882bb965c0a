"""GPU drop-in for the reference's ``parallel_multigrid`` module.

Same surface as core/parallel_multigrid.py of the reference checkout -- ``Multigrid(b0, x0, r0, h0,
epsilon, n, n_para=1)`` with ``.solve()``, ``.it``, ``.n``, ``.n_para`` and the level lists, plus
``gather_blocks``, ``split``, ``residual``, ``offsets`` -- with every array operation executed by
the HIP kernels of libmg_hip.so.  Two classes:

``Multigrid``        the reference's block layout, bit-identical results.  The reference runs four
                     MPI ranks, one per 2x2 block; here the four ranks are four ``Multigrid``
                     objects of ONE process that share a ``BlockWorld`` (the stand-in for
                     ``MPI.COMM_WORLD``), and their blocks are processed together on one GPU with
                     the reference's own schedule, halo-exchange order and stop rule
                     (parallel_multigrid.py:100-114, :117-138, :232-348).  Per-cycle errors,
                     iteration counts and the returned arrays equal the reference's bit for bit.
                     A per-rank driver written for mpiexec runs unchanged under
                     ``multigrid_amd.spmd.run`` (one thread per rank, ``spmd.MPI.COMM_WORLD``).
``GlobalMultigrid``  the same V-cycle on the undecomposed grid, backed by the row-slab engine
                     (multigrid_amd.engine): the fast path, 1..N GPUs.

SPMD in one process: ``solve()`` is collective.  The first ``solve()`` call of a round uploads the
level-0 arrays of all four ranks as they are at that moment (they alias the callers' numpy arrays,
as in the reference, parallel_multigrid.py:196-198), runs the solve for all four, writes x and r
back into every rank's arrays and returns the calling rank's error; the other three ranks' next
``solve()`` returns their own result of that same solve.  So the reference's per-rank program,
executed rank after rank, behaves as under mpiexec provided every rank has updated its right-hand
side before the first of them calls ``solve()``.
"""
import numpy as np
import torch

from . import multigrid_module as _mm
from ._lib import MgError, NORM_SLOTS, SOR_LEXICOGRAPHIC, check, lib
from .levels import BLOCK_OFFSETS, pitch
from .multigrid_module import residual, split  # noqa: F401  (re-exported like the reference, test/test.py:6-7)

__all__ = ["Multigrid", "GlobalMultigrid", "BlockWorld", "gather_blocks", "split", "residual", "offsets"]


def offsets(rank, width=1):
    """core/parallel_multigrid.py:141-149."""
    ofs = [{"j": 0, "i": 0}, {"j": 0, "i": -width}, {"j": -width, "i": 0}, {"j": -width, "i": -width}]
    return ofs[rank]


class _DeviceLevels:
    """Device storage of the reference's level arrays for all four ranks (rank-major batches)."""

    def __init__(self, n, n_para, device):
        self.dev = device
        z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=device)  # noqa: E731
        self.sb_m = [2 ** (n - i) + 2 for i in range(n_para + 1)]
        self.sb_ld = [pitch(m) for m in self.sb_m]
        self.x = [z(4, m, ld) for m, ld in zip(self.sb_m[:n_para], self.sb_ld)]
        self.r = [z(4, m, ld) for m, ld in zip(self.sb_m[:n_para], self.sb_ld)]
        self.b = [z(4, m, ld) for m, ld in zip(self.sb_m, self.sb_ld)]
        self.tmp = z(4, self.sb_m[0], self.sb_ld[0])
        self.wl_m = [2 ** (n - i) + 1 for i in range(n_para - 1, n)]
        self.wl_ld = [pitch(m) for m in self.wl_m]
        self.xw = [z(m, ld) for m, ld in zip(self.wl_m, self.wl_ld)]
        self.bw = [z(m, ld) for m, ld in zip(self.wl_m, self.wl_ld)]
        self.rw = [z(m, ld) for m, ld in zip(self.wl_m, self.wl_ld)]
        self.stage = z(8 * self.sb_m[0])
        self.norm = torch.zeros(NORM_SLOTS, dtype=torch.int64, device=device)


class BlockWorld:
    """The four block ranks of one solve (what MPI.COMM_WORLD is to the reference)."""

    size = 4

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise MgError("no GPU visible: multigrid_amd runs on MI355X only and has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.members = [None] * 4
        self._dl = None
        self._pending = [[] for _ in range(4)]  # per rank: (err, it) of the collective solves it has not been handed yet
        self.verbose = True

    # ---- membership
    def _register(self, mg):
        if self.members[mg.rank] is not None and self.members[mg.rank] is not mg:
            # a new solver object for this rank replaces the old one (a new problem on the same world)
            self._dl = None
            self._pending = [[] for _ in range(4)]
        self.members[mg.rank] = mg

    def _complete(self):
        ms = self.members
        if any(m is None for m in ms):
            missing = [k for k in range(4) if ms[k] is None]
            raise MgError(f"solve() is collective: ranks {missing} have no Multigrid object in this BlockWorld yet")
        a = ms[0]
        for m in ms[1:]:
            if (m.n, m.n_para, m.h0, m.epsilon) != (a.n, a.n_para, a.h0, a.epsilon):
                raise MgError("the four ranks disagree on (n, n_para, h0, epsilon)")
        return ms

    # ---- kernels on rank batches -------------------------------------------------------------
    def _s(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _exchange(self, v, m, ld):
        check(lib().mg_block_exchange(v.data_ptr(), m, ld, m * ld, self._dl.stage.data_ptr(), self._s()))

    def _smooth_parallel(self, b, x, h, r, m, ld, iterations):
        """smooth_parallel for the four ranks (core/parallel_multigrid.py:100-114): snapshot, exchange,
        iterations x (sweep into the snapshot, copy ALL of it back, exchange), residual."""
        L, s = lib(), self._s()
        tmp = self._dl.tmp
        check(L.mg_copy2d(tmp.data_ptr(), ld, x.data_ptr(), ld, 4 * m, m, s))
        self._exchange(x, m, ld)
        for _ in range(iterations):
            check(L.mg_jacobi(b.data_ptr(), x.data_ptr(), tmp.data_ptr(), m, m, ld, 1, m - 1, h, 1.0, 4, m * ld, 0, s))
            check(L.mg_copy2d(x.data_ptr(), ld, tmp.data_ptr(), ld, 4 * m, m, s))
            self._exchange(x, m, ld)
        check(L.mg_residual(r.data_ptr(), x.data_ptr(), b.data_ptr(), m, m, ld, 1, m - 1, h, None, 4, m * ld, 0, s))

    def _coarse(self, f, fm, fld, c, cm, cld):
        for rk in range(4):
            o = BLOCK_OFFSETS[rk]
            check(lib().mg_restrict(f[rk].data_ptr(), fm, fm, fld, c[rk].data_ptr(), cm, cm, cld, 1, cm - 1, o["i"],
                                    o["j"], 1, 0, 0, self._s()))

    def _interpolate(self, c, cm, cld, f, fm, fld):
        for rk in range(4):
            o = BLOCK_OFFSETS[rk]
            check(lib().mg_prolong_add(c[rk].data_ptr(), cm, cm, cld, f[rk].data_ptr(), fm, fm, fld, 0, fm, 0, fm - 1,
                                       o["i"], o["j"], 1, 0, 0, self._s()))

    def _gather(self, blk, m, ld, full, m0, ld0):
        """gather_blocks stitch, core/parallel_multigrid.py:135-138."""
        L, s, nx2 = lib(), self._s(), m - 1
        e = 8  # bytes per element
        check(L.mg_copy2d(full.data_ptr(), ld0, blk[0].data_ptr(), ld, nx2, nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * nx2, ld0, blk[1].data_ptr() + e * 2, ld, nx2, m0 - nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * nx2 * ld0, ld0, blk[2].data_ptr() + e * 2 * ld, ld, m0 - nx2, nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * (nx2 * ld0 + nx2), ld0, blk[3].data_ptr() + e * (2 * ld + 2), ld,
                          m0 - nx2, m0 - nx2, s))

    def _sor(self, k, iterations):
        D = self._dl
        m, ld = D.wl_m[k], D.wl_ld[k]
        check(lib().mg_sor(D.bw[k].data_ptr(), D.xw[k].data_ptr(), D.rw[k].data_ptr(), m, m, ld, self._h_wl[k],
                           iterations, 1, SOR_LEXICOGRAPHIC, 1, 0, self._s()))

    def _maxabs_per_rank(self, v, m, ld):
        import ctypes as C
        out = []
        L, s = lib(), self._s()
        for rk in range(4):
            val = C.c_double()
            check(L.mg_norm_reset(self._dl.norm.data_ptr(), s))
            check(L.mg_maxabs(v[rk].data_ptr(), m, m, ld, self._dl.norm.data_ptr(), s))
            check(L.mg_norm_read(self._dl.norm.data_ptr(), C.byref(val), s))
            out.append(val.value)
        return out

    # ---- the collective solve ------------------------------------------------------------------
    def solve_all(self):
        """Multigrid.solve() of the four ranks, core/parallel_multigrid.py:232-348.  Returns the four
        per-rank errors; sets .it on every member."""
        ms = self._complete()
        n, n_para, h0, eps = ms[0].n, ms[0].n_para, ms[0].h0, ms[0].epsilon
        with torch.cuda.device(self.device):
            if self._dl is None:
                self._dl = _DeviceLevels(n, n_para, self.device)
            D = self._dl
            m0, ld0 = D.sb_m[0], D.sb_ld[0]
            for rk, mg in enumerate(ms):  # level-0 arrays alias the callers' memory (:196-198)
                D.b[0][rk, :, :m0] = torch.from_numpy(np.ascontiguousarray(mg.b0)).to(self.device)
                D.x[0][rk, :, :m0] = torch.from_numpy(np.ascontiguousarray(mg.x0)).to(self.device)
                D.r[0][rk, :, :m0] = torch.from_numpy(np.ascontiguousarray(mg.r0)).to(self.device)
            h_sb = [h0 * 2 ** i for i in range(n_para)]
            self._h_wl = [h0 * 2 ** (i + 1) for i in range(n_para - 1, n)]
            n1, n2, L = 3, 6, lib()
            nwl = len(D.wl_m)
            it, status = 0, 0
            err = [0.0] * 4
            err_old = [100.0] * 4
            fail = [False] * 4
            self.err_history = []
            self._smooth_parallel(D.b[0], D.x[0], h_sb[0], D.r[0], m0, ld0, 1)  # :264-266
            while status < 4 and it < 100:  # :270
                # descent, sub-domains
                self._exchange(D.r[0], m0, ld0)
                self._coarse(D.r[0], m0, ld0, D.b[1], D.sb_m[1], D.sb_ld[1])
                for i in range(1, n_para):
                    m, ld = D.sb_m[i], D.sb_ld[i]
                    check(L.mg_zero(D.x[i].data_ptr(), D.x[i].numel(), self._s()))
                    self._smooth_parallel(D.b[i], D.x[i], h_sb[i], D.r[i], m, ld, n1)
                    self._exchange(D.r[i], m, ld)
                    self._coarse(D.r[i], m, ld, D.b[i + 1], D.sb_m[i + 1], D.sb_ld[i + 1])
                # whole domain (identical on every rank of the reference, computed once)
                check(L.mg_zero(D.xw[0].data_ptr(), D.xw[0].numel(), self._s()))
                self._gather(D.b[n_para], D.sb_m[n_para], D.sb_ld[n_para], D.bw[0], D.wl_m[0], D.wl_ld[0])
                self._sor(0, n1)
                for i in range(1, nwl):
                    check(L.mg_zero(D.xw[i].data_ptr(), D.xw[i].numel(), self._s()))
                    check(L.mg_restrict(D.rw[i - 1].data_ptr(), D.wl_m[i - 1], D.wl_m[i - 1], D.wl_ld[i - 1],
                                        D.bw[i].data_ptr(), D.wl_m[i], D.wl_m[i], D.wl_ld[i], 1, D.wl_m[i] - 1, 0, 0, 1,
                                        0, 0, self._s()))
                    self._sor(i, n1)
                for i in range(2, nwl):  # :302 -- never reaches x_wl[0]
                    c, f = nwl - i + 1, nwl - i
                    check(L.mg_prolong_add(D.xw[c].data_ptr(), D.wl_m[c], D.wl_m[c], D.wl_ld[c], D.xw[f].data_ptr(),
                                           D.wl_m[f], D.wl_m[f], D.wl_ld[f], 0, D.wl_m[f], 0, D.wl_m[f] - 1, 0, 0, 1, 0,
                                           0, self._s()))
                    self._sor(f, n2)
                # ascent, sub-domains
                ms_, lds_ = D.sb_m[n_para], D.sb_ld[n_para]
                for rk in range(4):
                    check(L.mg_split(D.xw[0].data_ptr(), D.wl_m[0], D.wl_m[0], D.wl_ld[0], D.b[n_para][rk].data_ptr(),
                                     ms_, lds_, rk, self._s()))
                self._interpolate(D.b[n_para], ms_, lds_, D.x[n_para - 1], D.sb_m[n_para - 1], D.sb_ld[n_para - 1])
                self._smooth_parallel(D.b[n_para - 1], D.x[n_para - 1], h_sb[n_para - 1], D.r[n_para - 1],
                                      D.sb_m[n_para - 1], D.sb_ld[n_para - 1], n2)
                for i in range(2, n_para + 1):
                    lc, lf = n_para - i + 1, n_para - i
                    self._interpolate(D.x[lc], D.sb_m[lc], D.sb_ld[lc], D.x[lf], D.sb_m[lf], D.sb_ld[lf])
                    self._smooth_parallel(D.b[lf], D.x[lf], h_sb[lf], D.r[lf], D.sb_m[lf], D.sb_ld[lf], n2)
                # convergence vote, every rank on its own block, halo lines included (:336-345)
                err = self._maxabs_per_rank(D.r[0], m0, ld0)
                self.err_history.append(list(err))
                status = 0
                for rk in range(4):
                    if err[rk] > err_old[rk] * 1.1:
                        if self.verbose:
                            print("Convergence compromised : r>r+1, set n1 higher")
                        fail[rk] = True
                    err_old[rk] = err[rk]
                    status += 0 if (err[rk] > eps and not fail[rk]) else 1
                it += 1
            for rk, mg in enumerate(ms):
                mg.x0[...] = D.x[0][rk, :, :m0].cpu().numpy()
                mg.r0[...] = D.r[0][rk, :, :m0].cpu().numpy()
                mg.it = it
        return err

    def level_array(self, name, rank, level):
        """Host copy of one level array of the last solve: name in x_sb, b_sb, r_sb, x_wl, b_wl, r_wl."""
        D = self._dl
        if name.endswith("_sb"):
            t = {"x_sb": D.x, "b_sb": D.b, "r_sb": D.r}[name][level]
            return t[rank, :, : D.sb_m[level]].cpu().numpy()
        t = {"x_wl": D.xw, "b_wl": D.bw, "r_wl": D.rw}[name][level]
        return t[:, : D.wl_m[level]].cpu().numpy()

    def _solve_for(self, rank):
        """solve() of one rank.  The four ranks of the reference call solve() together; here they call one after the
        other, so a collective solve runs when a rank asks for a result it has not been handed yet, and every rank's
        result of that round is queued until that rank asks.  Both calling patterns of the reference's drivers
        therefore run one collective solve per round: all ranks per round (test/test.py:76-82), or rank after rank
        with several solve() calls each (core/parallel_multigrid.py:384-415 executed rank by rank) -- in the second
        pattern a rank's arrays hold the LAST round's values by the time a later rank reads an earlier round's error."""
        if not self._pending[rank]:
            errs = self.solve_all()
            for k in range(4):
                self._pending[k].append((errs[k], self.members[k].it))
        err, it = self._pending[rank].pop(0)
        self.members[rank].it = it
        return err


_default_world = None
_current_rank = 0


def default_world():
    global _default_world
    if _default_world is None:
        _default_world = BlockWorld()
    return _default_world


def set_rank(rank):
    """Rank that ``Multigrid(...)`` takes when none is passed (the reference reads it from MPI)."""
    global _current_rank
    if rank not in (0, 1, 2, 3):
        raise ValueError("rank must be 0..3")
    _current_rank = rank


class _LevelList:
    """List-like view of one family of level arrays of a rank (x_sb, b_sb, ...): element 0 of the
    sub-domain families is the caller's own array, everything else is fetched from the device."""

    def __init__(self, mg, name, count, first=None):
        self._mg, self._name, self._count, self._first = mg, name, count, first

    def __len__(self):
        return self._count

    def __getitem__(self, i):
        if i < 0:
            i += self._count
        if not 0 <= i < self._count:
            raise IndexError(i)
        if i == 0 and self._first is not None:
            return self._first
        return self._mg.world.level_array(self._name, self._mg.rank, i)


class Multigrid:
    """Drop-in for the reference's ``Multigrid`` (core/parallel_multigrid.py:152-348) on the block
    layout.  Extra keyword arguments: ``rank`` (0..3, default: ``set_rank``'s value) and ``world``
    (default: a module-wide ``BlockWorld``)."""

    def __init__(self, b0, x0, r0, h0, epsilon, n, n_para=1, *, rank=None, world=None):
        for name, a in (("b0", b0), ("x0", x0), ("r0", r0)):
            if not isinstance(a, np.ndarray) or a.dtype != np.float64 or a.ndim != 2:
                raise TypeError(f"{name}: expected a 2-D float64 numpy array")
        _, self.nx = x0.shape
        self.n = n
        assert self.nx == 2 ** n + 2  # :157
        self.n_para = n_para
        assert n_para < n  # :159
        if b0.shape != x0.shape or r0.shape != x0.shape or x0.shape[0] != x0.shape[1]:
            raise MgError("b0, x0 and r0 must be square arrays of one shape")
        self.b0, self.x0, self.r0 = b0, x0, r0
        self.h0, self.epsilon = h0, epsilon
        # inside multigrid_amd.spmd.run (one thread per rank) the calling thread's rank and the run's world are the
        # defaults, so that the reference's constructor call needs no extra keyword
        from . import spmd
        ctx = spmd.current()
        if ctx is not None:
            rank = ctx[1] if rank is None else rank
            world = ctx[0].world if world is None else world
        self.rank = _current_rank if rank is None else rank
        if self.rank not in (0, 1, 2, 3):
            raise ValueError("rank must be 0..3")
        self.world = default_world() if world is None else world
        self.ofst = BLOCK_OFFSETS[self.rank]
        self.width_halo = 1
        self.it = 0
        self.h_sb = [h0 * 2 ** i for i in range(n_para)]
        self.h_wl = [h0 * 2 ** (i + 1) for i in range(n_para - 1, n)]
        self.x_sb = _LevelList(self, "x_sb", n_para, x0)
        self.b_sb = _LevelList(self, "b_sb", n_para + 1, b0)
        self.r_sb = _LevelList(self, "r_sb", n_para, r0)
        self.x_wl = _LevelList(self, "x_wl", n - n_para + 1)
        self.b_wl = _LevelList(self, "b_wl", n - n_para + 1)
        self.r_wl = _LevelList(self, "r_wl", n - n_para + 1)
        self.world._register(self)

    def solve(self):
        """Solve by repeating V-cycles; returns this rank's max|r| (halo lines included), as the
        reference does (:336, :348).  Collective over the four ranks of the world."""
        from . import spmd
        ctx = spmd.current()
        if ctx is not None and ctx[0].world is self.world and ctx[1] == self.rank:
            # the four ranks are four threads that call solve() together, as under mpiexec: rank 0's thread runs the
            # collective solve once all have arrived, every rank returns its own error
            return ctx[0].on_rank0(self.world.solve_all)[self.rank]
        return self.world._solve_for(self.rank)


def gather_blocks(comm, M_block, M_full):
    """Assemble the four sub-domain blocks into the whole-domain array
    (core/parallel_multigrid.py:117-138).  ``M_block`` is either a sequence of the four ranks' blocks
    or one rank's level-0 array (x0, b0 or r0 of a ``Multigrid`` registered in ``comm``), in which
    case the other ranks' corresponding arrays are looked up in the world ``comm``."""
    from . import spmd
    if isinstance(comm, spmd.ThreadComm) and not isinstance(M_block, (list, tuple)):
        # every rank passes its own block and gets the whole array, as after the reference's four broadcasts
        blocks = comm.allgather(M_block)
        with comm.gpu_lock:
            gather_blocks(None, blocks, M_full)
        comm.Barrier()  # nobody changes its block while another rank still copies from it
        return
    if isinstance(M_block, (list, tuple)):
        blocks = list(M_block)
    else:
        world = comm if isinstance(comm, BlockWorld) else default_world()
        blocks = None
        for attr in ("x0", "b0", "r0"):
            if any(m is not None and getattr(m, attr) is M_block for m in world.members):
                world._complete()
                blocks = [getattr(m, attr) for m in world.members]
        if blocks is None:
            raise MgError("gather_blocks: pass the four blocks, or a level-0 array of a registered Multigrid")
    if len(blocks) != 4:
        raise MgError("gather_blocks: four blocks expected")
    _, nx = blocks[0].shape
    nx2 = nx - 1
    _, nx0 = M_full.shape
    assert nx2 * 2 == nx0 + 1  # :122
    g = _mm._Args()
    full = g(M_full, "M_full", out=True)
    blk = [g(b, "M_block") for b in blocks]
    e = 8
    s = _mm._stream()
    L = lib()
    ld0, ld = _mm._ld(full), [_mm._ld(b) for b in blk]
    with torch.cuda.device(g.device):
        check(L.mg_copy2d(full.data_ptr(), ld0, blk[0].data_ptr(), ld[0], nx2, nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * nx2, ld0, blk[1].data_ptr() + e * 2, ld[1], nx2, nx0 - nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * nx2 * ld0, ld0, blk[2].data_ptr() + e * 2 * ld[2], ld[2], nx0 - nx2,
                          nx2, s))
        check(L.mg_copy2d(full.data_ptr() + e * (nx2 * ld0 + nx2), ld0, blk[3].data_ptr() + e * (2 * ld[3] + 2), ld[3],
                          nx0 - nx2, nx0 - nx2, s))
    g.finish()


class GlobalMultigrid:
    """The reference V-cycle on the undecomposed (2**(n+1)+1)^2 grid: same constructor arguments as
    ``Multigrid`` but b0/x0/r0 are the WHOLE grid, boundary included.  Backed by the row-slab engine
    (one GPU here; see multigrid_amd.engine.SlabMultigrid for multi-GPU use).  Level-0 arrays alias
    the caller's numpy arrays as in the reference: b0 and x0 are read at every solve(), x0 and r0
    are written back.  Results equal the block-layout ``Multigrid`` up to the reference's halo
    artefacts (see tests/test_gpu_engine.py for the measured deviations)."""

    def __init__(self, b0, x0, r0, h0, epsilon, n, n_para=1, **engine_kw):
        from .engine import SlabMultigrid
        N = 2 ** (n + 1) + 1
        if x0.shape != (N, N) or b0.shape != (N, N) or r0.shape != (N, N):
            raise MgError(f"GlobalMultigrid expects ({N}, {N}) arrays for n = {n}")
        assert n_para < n
        self.b0, self.x0, self.r0 = b0, x0, r0
        self.n, self.n_para, self.h0, self.epsilon = n, n_para, h0, epsilon
        self.engine = SlabMultigrid(N, N, n_para, h0, epsilon, world=1, rank=0, **engine_kw)
        self.it = 0

    def solve(self):
        e = self.engine
        dev = e.b0.device
        e.set_b_rows(torch.from_numpy(np.ascontiguousarray(self.b0)).to(dev))
        e.set_x_rows(torch.from_numpy(np.ascontiguousarray(self.x0)).to(dev))
        err = e.solve()
        self.x0[...] = e.local_rows(e.x0).cpu().numpy()
        self.r0[...] = e.local_rows(e.r0).cpu().numpy()
        self.it = e.it
        return err
