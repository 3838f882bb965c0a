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


class OracleOps:
    name = "oracle"
    supports_graphs = False

    def zeros(self, rows, ld):
        return torch.zeros((rows, ld), dtype=torch.float64)

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

    def zero_(self, t):
        t.zero_()

    def copy_(self, dst, src):
        dst.copy_(src)

    # ------------------------------------------------------------------ kernels
    def jacobi(self, b, x, xo, nx, j0, j1, h, finest=False):
        xs = np.ascontiguousarray(x.numpy()[j0 - 1 : j1 + 1, :nx])
        bs = np.ascontiguousarray(b.numpy()[j0 - 1 : j1 + 1, :nx])
        oracle.smooth_sweep_jacobi(bs, xs, xs.copy(), h)
        xo.numpy()[j0:j1, 1 : nx - 1] = xs[1:-1, 1:-1]

    def jacobi_from_zero(self, b, xo, nx, j0, j1, h, finest=False):
        bs = np.ascontiguousarray(b.numpy()[j0 - 1 : j1 + 1, :nx])
        xs = np.zeros_like(bs)
        oracle.smooth_sweep_jacobi(bs, xs, xs.copy(), h)
        xo.numpy()[j0:j1, 1 : nx - 1] = xs[1:-1, 1:-1]

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

    def restrict(self, f, fnx, c, cnx, J0, J1, oi, oj):
        f0 = 2 * (J0 - 1)
        fs = np.ascontiguousarray(f.numpy()[f0 : 2 * (J1 - 1) + oj + 2, :fnx])
        cs = np.zeros((J1 - J0 + 2, cnx))
        oracle.coarse(fs, cs, oi, oj)
        c.numpy()[J0:J1, 1 : cnx - 1] = cs[1:-1, 1:-1]

    def prolong_add(self, c, cnx, f, fnx, t0, t1, oi, oj):
        fc = np.ascontiguousarray(f.numpy()[:, :fnx])
        oracle.interpolate_add_to(np.ascontiguousarray(c.numpy()[:, :cnx]), fc, oi, oj)
        f.numpy()[t0:t1, 1 : fnx - 1] = fc[t0:t1, 1 : fnx - 1]

    supports_fused = True

    def smooth_fused(self, b, x_in, x_out, r, crs, nx, cnx, j0, j1, fix_lo, fix_hi, oj, sweeps, h, norm=None):
        """The fused pass as the sequence it stands for: prolong_add, `sweeps` Jacobi sweeps, residual
        (rows between the fixed rows are the interior; only rows j0..j1-1 are written back)."""
        cur = torch.zeros_like(x_out) if x_in is None else x_in.clone()
        if crs is not None:
            self.prolong_add(crs, cnx, cur, nx, fix_lo + 1, fix_hi, 0, oj)
        for _ in range(sweeps):
            nxt = cur.clone()
            self.jacobi(b, cur, nxt, nx, fix_lo + 1, fix_hi, h)
            cur = nxt
        x_out[j0:j1] = cur[j0:j1]
        if r is not None:
            self.residual(r, cur, b, nx, j0, j1, h, norm)

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

    def norm_tensor(self, slots):
        return slots.view(torch.float64)[:1].clone()
