"""TEST-ONLY stand-in for multigrid_amd.ops.HipOps that computes with the CPU oracle.

It lets the engine's host logic (slab partition, halo exchange, gather/scatter, schedule, stop
rule) run on a machine without a GPU, single-process or under torch.distributed/gloo.  It lives
under tests/ because only tests may use the oracle; the product never imports it.
"""
import contextlib

import numpy as np
import torch

import oracle

_SLOTS = 64


def _batched(fn):
    """Apply a per-grid method to every grid of a batch: tensor arguments with a leading batch dimension
    are sliced, everything else is passed through."""
    def wrapper(self, *args, **kw):
        nb = max([a.shape[0] for a in list(args) + list(kw.values()) if torch.is_tensor(a) and a.dim() == 3] or [0])
        if nb == 0:
            return fn(self, *args, **kw)
        pick = lambda a, k: a[k] if (torch.is_tensor(a) and a.dim() == 3) else a  # noqa: E731
        for k in range(nb):
            fn(self, *[pick(a, k) for a in args], **{n: pick(a, k) for n, a in kw.items()})
    return wrapper


class OracleOps:
    name = "oracle"
    supports_graphs = False

    def zeros(self, rows, ld, batch=None):
        return torch.zeros((rows, ld) if batch is None else (batch, rows, ld), dtype=torch.float64)

    def norm_slots(self):
        return torch.zeros(_SLOTS, dtype=torch.int64)

    def ctx(self):
        return contextlib.nullcontext()

    def begin(self):
        pass

    def end(self):
        pass

    def synchronize(self):
        pass

    def mark(self):
        return None

    def side(self, after=None):
        return contextlib.nullcontext()

    def join_side(self):
        pass

    def comm_side(self, after=None):
        return contextlib.nullcontext()

    def comm_join(self):
        pass

    def zero_(self, t):
        t.zero_()

    def copy_(self, dst, src):
        dst.copy_(src)

    # ------------------------------------------------------------------ kernels
    @_batched
    def jacobi(self, b, x, xo, nx, j0, j1, h, finest=False, omega=1.0):
        xs = np.ascontiguousarray(x.numpy()[j0 - 1 : j1 + 1, :nx])
        bs = np.ascontiguousarray(b.numpy()[j0 - 1 : j1 + 1, :nx])
        oracle.smooth_sweep_jacobi(bs, xs, xs.copy(), h, omega)
        xo.numpy()[j0:j1, 1 : nx - 1] = xs[1:-1, 1:-1]

    @_batched
    def jacobi_from_zero(self, b, xo, nx, j0, j1, h, finest=False, omega=1.0):
        bs = np.ascontiguousarray(b.numpy()[j0 - 1 : j1 + 1, :nx])
        xs = np.zeros_like(bs)
        oracle.smooth_sweep_jacobi(bs, xs, xs.copy(), h, omega)
        xo.numpy()[j0:j1, 1 : nx - 1] = xs[1:-1, 1:-1]

    @_batched
    def residual(self, r, x, b, nx, j0, j1, h, norm=None, finest=False):
        xs = np.ascontiguousarray(x.numpy()[j0 - 1 : j1 + 1, :nx])
        bs = np.ascontiguousarray(b.numpy()[j0 - 1 : j1 + 1, :nx])
        rs = np.zeros_like(xs)
        oracle.residual(rs, xs, bs, h)
        r.numpy()[j0:j1, 1 : nx - 1] = rs[1:-1, 1:-1]
        if norm is not None:
            m = np.abs(rs[1:-1, 1:-1])
            v = np.nan if np.isnan(m).any() else m.max()
            cur = norm.view(torch.float64)
            cur[0] = v if (np.isnan(v) or v > cur[0].item()) else cur[0].item()

    # The oracle's transfer functions take the reference's row offsets 0 / -1.  A slab with an H-row
    # halo uses offset -H: the same operator on row-shifted windows of the arrays.
    @_batched
    def restrict(self, f, fnx, c, cnx, J0, J1, oi, oj):
        o2 = 0 if oj % 2 == 0 else -1
        f0 = 2 * (J0 - 1) + oj - o2  # first fine row of the window: its local offset becomes o2
        fs = np.ascontiguousarray(f.numpy()[f0 : 2 * (J1 - 1) + oj + 2, :fnx])
        cs = np.zeros((J1 - J0 + 2, cnx))
        oracle.coarse(fs, cs, oi, o2)
        c.numpy()[J0:J1, 1 : cnx - 1] = cs[1:-1, 1:-1]

    @_batched
    def prolong_add(self, c, cnx, f, fnx, t0, t1, oi, oj):
        fc = np.ascontiguousarray(f.numpy()[:, :fnx])
        if oj in (0, -1):
            oracle.interpolate_add_to(np.ascontiguousarray(c.numpy()[:, :cnx]), fc, oi, oj)
        else:
            # fine row t <-> coarse row (t - oj) / 2.  Drop the first -oj/2 - 1 coarse rows and put two
            # spare rows in front of the fine array: then fine row t+2 <-> coarse row' (t+2)/2, offset 0,
            # and fine row 0 is still reached (the oracle's passes start at fine row 1).
            assert oj % 2 == 0
            s = -oj // 2 - 1
            fp = np.zeros((fc.shape[0] + 2, fnx))
            fp[2:] = fc
            n = min(c.shape[0] - s, (fp.shape[0] + 2) // 2)
            oracle.interpolate_add_to(np.ascontiguousarray(c.numpy()[s : s + n, :cnx]), fp, oi, 0)
            fc = fp[2:]
        f.numpy()[t0:t1, 1 : fnx - 1] = fc[t0:t1, 1 : fnx - 1]

    supports_fused = True
    supports_prolong_after = True
    supports_also_rows = True

    def smooth_fused(self, b, x_in, x_out, r, crs, nx, cnx, j0, j1, fix_lo, fix_hi, oj, sweeps, h, norm=None,
                     coarse_b=None, cbnx=0, T0=0, T1=0, prolong_after=0, omega=1.0, also_rows=None):
        """also_rows = (j2, j3, T2, T3): a second row range of the same pass, here simply a second pass."""
        self._smooth_fused(b, x_in, x_out, r, crs, nx, cnx, j0, j1, fix_lo, fix_hi, oj, sweeps, h, norm, coarse_b, cbnx, T0, T1,
                           prolong_after, omega)
        if also_rows is not None:
            j2, j3, T2, T3 = also_rows
            self.also_rows_calls = getattr(self, "also_rows_calls", 0) + 1
            self._smooth_fused(b, x_in, x_out, r, crs, nx, cnx, j2, j3, fix_lo, fix_hi, oj, sweeps, h, norm, coarse_b, cbnx, T2, T3,
                               prolong_after, omega)

    @_batched
    def _smooth_fused(self, b, x_in, x_out, r, crs, nx, cnx, j0, j1, fix_lo, fix_hi, oj, sweeps, h, norm=None,
                      coarse_b=None, cbnx=0, T0=0, T1=0, prolong_after=0, omega=1.0):
        """The fused pass as the sequence it stands for: prolong_add (before the first sweep, or after
        `prolong_after` sweeps from zero), `sweeps` Jacobi sweeps, residual (rows between the fixed rows
        are the interior; only rows j0..j1-1 are written back)."""
        ny = b.shape[0]
        lo, hi = max(fix_lo + 1, 1), min(fix_hi, ny - 1)  # rows a sweep may update; what lies beyond a
        cur = torch.zeros_like(b) if x_in is None else x_in.clone()  # non-fixed end is halo
        if prolong_after:
            assert x_in is None and crs is not None
        for k in range(sweeps + 1):
            if crs is not None and k == prolong_after:
                self.prolong_add(crs, cnx, cur, nx, max(fix_lo + 1, 0), min(fix_hi, ny), 0, oj)
            if k == sweeps:
                break
            nxt = cur.clone()
            self.jacobi(b, cur, nxt, nx, lo, hi, h, omega=omega)
            cur = nxt
        if x_out is not None:
            x_out[j0:j1] = cur[j0:j1]
        if r is not None:
            self.residual(r, cur, b, nx, j0, j1, h, norm)
        if coarse_b is not None:
            # the residual on every row a sweep may touch (halo rows included), Dirichlet cells as r holds them
            rt = r.clone() if r is not None else torch.zeros_like(b)
            self.residual(rt, cur, b, nx, lo, hi, h)
            if r is None and norm is not None:  # the norm counts the produced rows only
                self.residual(torch.zeros_like(b), cur, b, nx, j0, j1, h, norm)
            self.restrict(rt, nx, coarse_b, cbnx, T0, T1, 0, oj)

    @_batched
    def sor(self, b, x, r, nx, h, iterations, ordering):
        xs = np.ascontiguousarray(x.numpy()[:, :nx])
        bs = np.ascontiguousarray(b.numpy()[:, :nx])
        rs = np.zeros_like(xs)
        (oracle.smooth_altern if ordering == "lexicographic" else oracle.smooth_altern_rb)(bs, xs, h, rs, iterations)
        x.numpy()[:, :nx] = xs
        if r is not None:
            r.numpy()[1:-1, 1 : nx - 1] = rs[1:-1, 1:-1]

    def norm_reset(self, slots):
        slots.zero_()

    def norm_read(self, slots):
        return float(slots.view(torch.float64)[0].item())

    def norm_read_at(self, slots, mark):
        return self.norm_read(slots)
