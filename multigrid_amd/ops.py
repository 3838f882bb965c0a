"""The kernel set the V-cycle engine is written against, bound to libmg_hip.so.

`HipOps` is a thin object: each method takes full-pitch 2-D float64 CUDA tensors (rows x ld) plus
the logical column count, and enqueues one call of the C ABI (include/mg_hip.h) on `stream`.
The engine only ever talks to an object with this interface.
"""
import contextlib
import ctypes as C

import os

import numpy as np
import torch

from . import _lib
from ._lib import check


class HipOps:
    name = "hip"

    def __init__(self, device=None):
        if not torch.cuda.is_available():
            raise _lib.MgError("no GPU visible: multigrid_amd runs on MI355X only and has no CPU path")
        self.L = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        # a stream of its own: the null stream cannot be captured into a hipGraph
        self.stream = torch.cuda.Stream(self.device)
        self._side, self._on_side, self._side_pending = None, False, False
        self._copy = None
        self._comm, self._on_comm, self._comm_done = None, False, None
        self._cap_swap = None  # (attribute, stream) that a graph capture's own stream stands in for

    # ---- memory / stream plumbing (torch) -------------------------------------------------
    def zeros(self, rows, ld, batch=None):
        return torch.zeros((rows, ld) if batch is None else (batch, rows, ld), dtype=torch.float64, device=self.device)

    supports_placement = True  # SlabMultigrid(tune_placement=True) may time placements of its arrays

    def norm_slots(self):
        return torch.zeros(_lib.NORM_SLOTS, dtype=torch.int64, device=self.device)

    @property
    def _s(self):
        """The stream kernels are enqueued on: the communication stream inside comm_side(), the side stream inside
        side(), else the main one."""
        return (self._comm if self._on_comm else self._side if self._on_side else self.stream).cuda_stream

    # Work that nothing later in the cycle reads can run beside the main sequence: side() routes the
    # calls made inside it to a second stream that starts after everything enqueued so far, and
    # join_side() makes the main stream wait for it.  Both are event record/wait pairs, so they
    # also work while the main stream is being captured into a graph (the side stream joins the capture
    # and must be joined back before the capture ends).
    def mark(self):
        """A point of the main sequence that side(after=...) can start from later."""
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return ev

    @contextlib.contextmanager
    def side(self, after=None):
        """after: a mark() -- the side work starts once the main sequence has reached that point,
        though it is enqueued (or captured) later; default: everything enqueued so far."""
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        ev = after if after is not None else self.mark()
        self._side.wait_event(ev)
        self._on_side, self._side_pending = True, True
        try:
            yield
        finally:
            self._on_side = False

    def join_side(self):
        if self._side_pending:
            ev = torch.cuda.Event()
            ev.record(self._side)
            self.stream.wait_event(ev)
            self._side_pending = False

    # An exchange whose result is not needed at once can run beside the kernels that follow it: inside
    # comm_side() the context torch.distributed calls are issued in (ctx()) is a communication stream
    # that starts at a mark() of the main sequence (default: after everything enqueued so far); the
    # main sequence goes on without waiting until comm_join().  Kernels launched inside comm_side() go to
    # that stream too, behind the exchange: the rows of a pass that need the halo run there, beside the
    # interior rows on the main stream.
    @contextlib.contextmanager
    def comm_side(self, after=None):
        if self._comm is None:
            self._comm = torch.cuda.Stream(self.device)
        self._comm.wait_event(after if after is not None else self.mark())
        self._on_comm = True
        try:
            yield
        finally:
            self._on_comm = False
            self._comm_done = torch.cuda.Event()
            self._comm_done.record(self._comm)

    def comm_join(self):
        if self._comm_done is not None:
            self.stream.wait_event(self._comm_done)
            self._comm_done = None

    def ctx(self):
        """Context in which torch ops and torch.distributed calls are ordered with the kernels."""
        return torch.cuda.stream(self._comm if self._on_comm else self.stream)

    def begin(self):
        """Order this object's stream after whatever the caller enqueued on torch's current stream."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def end(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def zero_(self, t):
        """t[:] = 0 for a tensor that covers whole rows of its allocation (contiguous memory)."""
        assert t.is_contiguous()
        check(self.L.mg_zero(t.data_ptr(), t.numel(), self._s))

    def copy_(self, dst, src):
        assert dst.shape == src.shape and dst.stride(0) == src.stride(0)
        check(self.L.mg_copy2d(dst.data_ptr(), dst.stride(0), src.data_ptr(), src.stride(0), dst.shape[0],
                               dst.shape[1], self._s))

    def synchronize(self):
        self.stream.synchronize()

    # ---- kernels ---------------------------------------------------------------------------
    # Arrays are (rows x ld) tensors, or (batch x rows x ld) for a batch of independent grids: every
    # array of a call then has the same leading batch dimension and one launch covers all grids.
    @staticmethod
    def _nb(t):
        """(nbatch, batch stride, rows, ld) of a 2-D or 3-D level array"""
        if t.dim() == 3:
            return t.shape[0], t.stride(0), t.shape[1], t.stride(1)
        return 1, 0, t.shape[0], t.stride(0)

    @staticmethod
    def _same_layout(*ts):
        """the entry points with ONE batch stride and leading dimension need arrays laid out alike"""
        ts = [t for t in ts if t is not None]
        if any(t.stride() != ts[0].stride() or t.shape != ts[0].shape for t in ts[1:]):
            raise _lib.MgError("arrays of one launch must share shape and strides: " +
                               ", ".join(f"{tuple(t.shape)}/{t.stride()}" for t in ts))

    def jacobi(self, b, x, xo, nx, j0, j1, h, finest=False, omega=1.0):
        self._same_layout(b, x, xo)
        nb, bs, ny, ld = self._nb(x)
        check(self.L.mg_jacobi(b.data_ptr(), x.data_ptr(), xo.data_ptr(), ny, nx, ld, j0, j1, h, omega, nb, bs,
                               _lib.FLAG_FINEST if finest else 0, self._s))

    def jacobi_from_zero(self, b, xo, nx, j0, j1, h, finest=False, omega=1.0):
        self._same_layout(b, xo)
        nb, bs, ny, ld = self._nb(xo)
        check(self.L.mg_jacobi_from_zero(b.data_ptr(), xo.data_ptr(), ny, nx, ld, j0, j1, h, omega, nb, bs,
                                         _lib.FLAG_FINEST if finest else 0, self._s))

    def residual(self, r, x, b, nx, j0, j1, h, norm=None, finest=False):
        self._same_layout(r, x, b)
        nb, bs, ny, ld = self._nb(x)
        check(self.L.mg_residual(r.data_ptr(), x.data_ptr(), b.data_ptr(), ny, nx, ld, j0, j1, h,
                                 norm.data_ptr() if norm is not None else None, nb, bs,
                                 _lib.FLAG_FINEST if finest else 0, self._s))

    def restrict(self, f, fnx, c, cnx, J0, J1, oi, oj):
        nb, fbs, fny, fld = self._nb(f)
        _, cbs, cny, cld = self._nb(c)
        check(self.L.mg_restrict(f.data_ptr(), fny, fnx, fld, c.data_ptr(), cny, cnx, cld, J0, J1, oi, oj, nb, fbs, cbs,
                                 self._s))

    def prolong_add(self, c, cnx, f, fnx, t0, t1, oi, oj):
        nb, fbs, fny, fld = self._nb(f)
        _, cbs, cny, cld = self._nb(c)
        check(self.L.mg_prolong_add(c.data_ptr(), cny, cnx, cld, f.data_ptr(), fny, fnx, fld, t0, t1, 1, fnx - 2, oi, oj,
                                    nb, cbs, fbs, self._s))

    supports_fused = True
    supports_prolong_after = True

    supports_also_rows = True

    def smooth_fused(self, b, x_in, x_out, r, crs, nx, cnx, j0, j1, fix_lo, fix_hi, oj, sweeps, h, norm=None,
                     coarse_b=None, cbnx=0, T0=0, T1=0, prolong_after=0, omega=1.0, also_rows=None):
        """[x_in += P(crs)] -> `sweeps` Jacobi sweeps -> [residual] -> [restriction into coarse_b];
        x_in None = start from zero; x_out None (with coarse_b) = x is not stored; prolong_after = n:
        the prolongation is added after the first n sweeps (from zero) instead of before the first;
        omega < 1: damped sweeps, x + omega * (t - x).  also_rows = (j2, j3, T2, T3): the same launch
        also produces rows j2 .. j3-1 (and restricts into coarse rows T2 .. T3-1)."""
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        self._same_layout(b, x_in, x_out, r)
        nb, bs, ny, ld = self._nb(b)
        _, cs, cny, cld = self._nb(crs) if crs is not None else (1, 0, 0, 0)
        common = (p(b), p(x_in), p(x_out), p(r), p(crs), ny, nx, ld, cny, cnx, cld, j0, j1, fix_lo, fix_hi, oj, sweeps)
        if coarse_b is not None:
            _, cbs, cbny, cbld = self._nb(coarse_b)
            assert prolong_after == 0
            # a (batch, slots) accumulator: every problem of the batch has its own max-norm
            nstride = norm.stride(0) if (norm is not None and norm.dim() == 2) else 0
        if also_rows is not None:  # consumed by the call that follows, whatever it returns
            check(self.L.mg_fused_also_rows(*also_rows, ny, self._s))
        if coarse_b is None:
            check(self.L.mg_smooth_fused(*common, prolong_after, h, omega, p(norm), nb, bs, cs, self._s))
        else:
            check(self.L.mg_smooth_fused_restrict(*common, h, omega, p(norm), nstride, p(coarse_b), cbny, cbnx, cbld, T0, T1, nb,
                                                  bs, cs, cbs, self._s))

    def gauss_seidel_rb(self, b, x, nx, j0, j1, h, row_offset=0, colour=None):
        """One red-black Gauss-Seidel sweep in place: first the cells with even (global row + column), then the odd
        ones; colour = 0 / 1: only that half of it.  row_offset: local row of global row 0 (the halo depth), so that
        the colours are those of the grid."""
        self._same_layout(b, x)
        nb, bs, ny, ld = self._nb(x)
        for colour in ((0, 1) if colour is None else (colour,)):
            check(self.L.mg_gauss_seidel_rb(b.data_ptr(), x.data_ptr(), ny, nx, ld, j0, j1, h, (colour + row_offset) & 1, nb, bs,
                                            self._s))

    def sor(self, b, x, r, nx, h, iterations, ordering):
        self._same_layout(b, x, r)
        nb, bs, ny, ld = self._nb(x)
        check(self.L.mg_sor(b.data_ptr(), x.data_ptr(), r.data_ptr() if r is not None else None, ny, nx, ld, h,
                            iterations, 1, {"lexicographic": _lib.SOR_LEXICOGRAPHIC, "redblack": _lib.SOR_REDBLACK}[ordering],
                            nb, bs, self._s))

    def norm_slots_per_problem(self, batch):
        """One accumulator per problem of a batch: a (batch, slots) tensor."""
        return torch.zeros((batch, _lib.NORM_SLOTS), dtype=torch.int64, device=self.device)

    def norm_reset(self, slots):
        if slots.dim() == 2:
            check(self.L.mg_zero(slots.data_ptr(), slots.numel(), self._s))  # 8-byte words, like doubles
        else:
            check(self.L.mg_norm_reset(slots.data_ptr(), self._s))

    def norm_read_per_problem(self, slots):
        """max|r| of every problem from a (batch, slots) accumulator (a float64 numpy array); waits for the stream.
        The accumulator holds bit patterns of |r|, whose integer order is the order of the values (NaN on top)."""
        with self.ctx():
            return slots.max(dim=1).values.view(torch.float64).cpu().numpy()

    def norm_read(self, slots):
        """max of the accumulator; waits for the stream."""
        if slots.dim() == 2:
            per = self.norm_read_per_problem(slots)
            return float("nan") if np.isnan(per).any() else float(per.max())
        out = C.c_double()
        check(self.L.mg_norm_read(slots.data_ptr(), C.byref(out), self._s))
        return out.value

    def norm_read_at(self, slots, mark):
        """max of the accumulator as it stands at a mark() of the main sequence: read on a stream of its
        own, so work enqueued after the mark keeps running while the host waits."""
        if self._copy is None:
            self._copy = torch.cuda.Stream(self.device)
        self._copy.wait_event(mark)
        if slots.dim() == 2:
            with torch.cuda.stream(self._copy):
                per = slots.max(dim=1).values.view(torch.float64).cpu().numpy()
            return float("nan") if np.isnan(per).any() else float(per.max())
        out = C.c_double()
        check(self.L.mg_norm_read(slots.data_ptr(), C.byref(out), self._copy.cuda_stream))
        return out.value

    # ---- hipGraph ----------------------------------------------------------------------------
    supports_graphs = True

    def graph_begin(self):
        """Start recording the calls that follow into a graph instead of running them.  They are recorded on a
        stream the library makes for this one capture (mg_graph_begin_own), which stands in for the current one
        (main or side) until graph_end(): in a process that holds an RCCL communicator, torch's watchdog thread
        polls the end events of outstanding collectives, and on this HIP runtime such a poll fails with
        hipErrorCapturedEvent -- and takes the process down -- while a stream that has waited for one of those
        events is capturing (tools/rccl_capture_probe.py).  A graph can be launched on any stream."""
        assert self._cap_swap is None, "graph captures do not nest"
        raw = C.c_void_p()
        check(self.L.mg_graph_begin_own(C.byref(raw)))
        cap = torch.cuda.ExternalStream(raw.value, device=self.device)
        if self._on_side:
            self._cap_swap = ("_side", self._side)
            self._side = cap
        else:
            self._cap_swap = ("stream", self.stream)
            self.stream = cap

    def graph_end(self):
        g = C.c_void_p()
        try:
            check(self.L.mg_graph_end(self._s, C.byref(g)))  # releases the capture's stream as well
        finally:
            name, stream = self._cap_swap
            setattr(self, name, stream)
            self._cap_swap = None
        return g

    def graph_abort(self):
        """End a capture that failed half-way: the recording stops, whatever graph came out of it is destroyed, and
        the stream the capture's own stream stood in for is back in place.  The error that interrupted the capture
        is the caller's to re-raise."""
        if self._cap_swap is None:
            return
        g = C.c_void_p()
        try:
            if self.L.mg_graph_end(self._s, C.byref(g)) == 0 and g:
                self.L.mg_graph_destroy(g)
        finally:
            name, stream = self._cap_swap
            setattr(self, name, stream)
            self._cap_swap = None

    @contextlib.contextmanager
    def capture(self):
        """with ops.capture() as cap: ...calls to record...; cap.graph is the executable graph afterwards.  An exception
        inside the block ends the capture cleanly (graph_abort) before it propagates."""
        class _Cap:
            graph = None
        cap = _Cap()
        self.graph_begin()
        try:
            yield cap
        except BaseException:
            self.graph_abort()
            raise
        cap.graph = self.graph_end()

    def graph_launch(self, g):
        check(self.L.mg_graph_launch(g, self._s))

    def graph_destroy(self, g):
        check(self.L.mg_graph_destroy(g))
