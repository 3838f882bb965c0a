#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference's own Python.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where the reference checkout is
mounted read-only at /root/reference; it never runs on the GPU box and nothing here is
imported by the product package.

How the reference is executed
-----------------------------
* ``core/numba_modules.py`` only uses numba for the ``@cc.export`` decorator of
  ``numba.pycc.CC`` (numba_modules.py:3-5, :183-184).  numba is not installed here, so a
  module object named ``numba.pycc`` is registered whose ``CC.export`` returns the identity
  decorator: the eight kernels then run as the plain Python loops they are written as.
  IEEE-754 double arithmetic is the same as in the numba build (no fastmath is requested
  there); the one spelling that could differ, ``h ** 2`` (libm ``pow`` here, ``h*h`` in
  numba), is asserted to be bit-identical for every ``h`` used below.
* ``core/parallel_multigrid.py`` needs ``mpi4py.MPI`` (persistent send/recv, Bcast,
  allreduce, reduce).  mpi4py/mpiexec are not installed, so an in-process stand-in runs the
  four ranks as four threads and moves the same bytes in the same order (FIFO per
  (source, destination, tag), which is MPI's non-overtaking rule).
* The reference records no history, so two observation hooks are used, neither of which
  changes a value: the stand-in ``allreduce`` (parallel_multigrid.py:342-345 calls it once
  per V-cycle) evaluates ``np.amax(np.abs(r_sb[0]))`` -- the expression of :336 -- on the
  calling rank, and a wrapper around ``coarse`` notes ``max|r_sb[0]|`` when it is called
  for level 0 (:273, i.e. right after the halo exchange of ``r``).

Outputs (all small, committed): tests/golden/kernels.npz, layout.json, global_*.npz, solve_*.npz.
"""
import json
import os
import queue
import sys
import threading
import types
from collections import defaultdict
from concurrent.futures import ProcessPoolExecutor

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np

REF_CORE = "/root/reference/core"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# --------------------------------------------------------------------------- stand-ins


def _install_numba_stub():
    class CC:
        def __init__(self, name):
            self.name = name

        def export(self, name, sig):
            return lambda f: f

        def compile(self):
            pass

    numba = types.ModuleType("numba")
    pycc = types.ModuleType("numba.pycc")
    pycc.CC = CC
    numba.pycc = pycc
    sys.modules["numba"] = numba
    sys.modules["numba.pycc"] = pycc


_tls = threading.local()


class _Req:
    def __init__(self, comm, kind, buf, me, peer, tag):
        self.comm, self.kind, self.buf, self.me, self.peer, self.tag = comm, kind, buf, me, peer, tag


class _Comm:
    """Four ranks = four threads of one process."""

    def __init__(self, size):
        self.size = size
        self.barrier = threading.Barrier(size)
        self.queues = defaultdict(queue.Queue)
        self.slots = [None] * size
        self.bcast_slot = None
        self.allreduce_hook = None  # called as hook(rank) before the vote is combined

    def Get_rank(self):
        return _tls.rank

    def Get_size(self):
        return self.size

    def Send_init(self, buf, dest, tag=0):
        return _Req(self, "s", buf, _tls.rank, dest, tag)

    def Recv_init(self, buf, source, tag=0):
        return _Req(self, "r", buf, _tls.rank, source, tag)

    def Bcast(self, buf, root=0):
        if _tls.rank == root:
            self.bcast_slot = buf
        self.barrier.wait()
        if _tls.rank != root:
            buf[...] = self.bcast_slot
        self.barrier.wait()

    def _combine(self, val, op):
        self.slots[_tls.rank] = val
        self.barrier.wait()
        res = max(self.slots) if op == "MAX" else sum(self.slots)
        self.barrier.wait()
        return res

    def allreduce(self, val, op="SUM"):
        if self.allreduce_hook is not None:
            self.allreduce_hook(_tls.rank)
        return self._combine(val, op)

    def reduce(self, val, op="SUM", root=0):
        res = self._combine(val, op)
        return res if _tls.rank == root else None

    def Barrier(self):
        self.barrier.wait()


class _Prequest:
    @staticmethod
    def Startall(reqs):
        for r in reqs:
            if r.kind == "s":
                r.comm.queues[(r.me, r.peer, r.tag)].put(np.array(r.buf, copy=True))

    @staticmethod
    def Waitall(reqs):
        for r in reqs:
            if r.kind == "r":
                r.buf[...] = r.comm.queues[(r.peer, r.me, r.tag)].get(timeout=600)


def _install_mpi_stub():
    mpi4py = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = _Comm(4)
    MPI.Prequest = _Prequest
    MPI.MAX = "MAX"
    MPI.SUM = "SUM"
    mpi4py.MPI = MPI
    sys.modules["mpi4py"] = mpi4py
    sys.modules["mpi4py.MPI"] = MPI
    return MPI


_REF = None


def _load_reference():
    global _REF
    if _REF is None:
        _REF = _load_reference_once()
    return _REF


def _load_reference_once():
    _install_numba_stub()
    MPI = _install_mpi_stub()
    if REF_CORE not in sys.path:
        sys.path.insert(0, REF_CORE)
    import numba_modules as nm

    mm = types.ModuleType("multigrid_module")  # what `numba_modules.py` compiles to
    for k in ("split", "laplacian", "residual", "smooth_sweep_jacobi", "smooth",
              "smooth_altern", "coarse", "interpolate_add_to"):
        setattr(mm, k, getattr(nm, k))
    sys.modules["multigrid_module"] = mm
    import parallel_multigrid as pm

    return nm, pm, MPI


def _check_h(h):
    assert h ** 2 == h * h and 1 / (h ** 2) == 1 / (h * h), h
    return h


# --------------------------------------------------------------------------- kernel vectors


def gen_kernels():
    nm, _, _ = _load_reference()
    rng = np.random.default_rng(0)
    out = {}
    meta = []

    def rnd(shape):
        return rng.standard_normal(shape)

    # residual(r, a, b, h): writes the interior of r only
    for k, (shape, h) in enumerate([((10, 10), 0.3), ((18, 18), 20 / 64), ((7, 12), 0.7), ((34, 34), 0.123)]):
        _check_h(h)
        a, b, r0 = rnd(shape), rnd(shape), rnd(shape)
        r = r0.copy()
        nm.residual(r, a, b, h)
        out.update({f"residual{k}_a": a, f"residual{k}_b": b, f"residual{k}_r0": r0, f"residual{k}_r": r})
        meta.append({"kernel": "residual", "id": k, "h": h})

    # laplacian(a, lplc, h)
    for k, (shape, h) in enumerate([((10, 10), 0.3), ((9, 17), 0.625)]):
        _check_h(h)
        a, l0 = rnd(shape), rnd(shape)
        l = l0.copy()
        nm.laplacian(a, l, h)
        out.update({f"laplacian{k}_a": a, f"laplacian{k}_l0": l0, f"laplacian{k}_l": l})
        meta.append({"kernel": "laplacian", "id": k, "h": h})

    # smooth_sweep_jacobi(b, a, a_temp, h): a_temp pre-filled with a as smooth_parallel does
    # (parallel_multigrid.py:105-106); one case with an unrelated a_temp shows the full copy-back.
    for k, (shape, h, same) in enumerate([((10, 10), 0.3, True), ((18, 18), 20 / 64, True),
                                          ((7, 12), 0.7, False), ((34, 34), 0.123, True)]):
        _check_h(h)
        a0, b = rnd(shape), rnd(shape)
        t0 = a0.copy() if same else rnd(shape)
        a, t = a0.copy(), t0.copy()
        nm.smooth_sweep_jacobi(b, a, t, h)
        out.update({f"jacobi{k}_b": b, f"jacobi{k}_a0": a0, f"jacobi{k}_t0": t0, f"jacobi{k}_a": a, f"jacobi{k}_t": t})
        meta.append({"kernel": "smooth_sweep_jacobi", "id": k, "h": h})

    # smooth / smooth_altern(b, x, h, r, iterations)
    k = 0
    for name in ("smooth", "smooth_altern"):
        for shape, h, its in [((10, 10), 0.3, 1), ((18, 18), 20 / 64, 3), ((9, 17), 0.7, 6),
                              ((33, 33), 0.625, 3), ((5, 5), 1.0, 2), ((3, 3), 0.3, 3)]:
            _check_h(h)
            b, x0, r0 = rnd(shape), rnd(shape), rnd(shape)
            x, r = x0.copy(), r0.copy()
            getattr(nm, name)(b, x, h, r, its)
            out.update({f"sor{k}_b": b, f"sor{k}_x0": x0, f"sor{k}_r0": r0, f"sor{k}_x": x, f"sor{k}_r": r})
            meta.append({"kernel": name, "id": k, "h": h, "iterations": its})
            k += 1

    # coarse(a, a_crs, ofst_i, ofst_j) / interpolate_add_to(a, a_new, ofst_i, ofst_j)
    k = 0
    cases = [((18, 18), (10, 10), oi, oj) for oi in (0, -1) for oj in (0, -1)]
    cases += [((34, 34), (18, 18), 0, -1), ((17, 17), (9, 9), 0, 0), ((9, 9), (5, 5), 0, 0),
              ((5, 5), (3, 3), 0, 0), ((10, 34), (6, 18), -1, 0)]
    for fshape, cshape, oi, oj in cases:
        f, c0 = rnd(fshape), rnd(cshape)
        c = c0.copy()
        nm.coarse(f, c, oi, oj)
        out.update({f"coarse{k}_f": f, f"coarse{k}_c0": c0, f"coarse{k}_c": c})
        meta.append({"kernel": "coarse", "id": k, "ofst_i": oi, "ofst_j": oj})
        cc, f0 = rnd(cshape), rnd(fshape)
        ff = f0.copy()
        nm.interpolate_add_to(cc, ff, oi, oj)
        out.update({f"interp{k}_c": cc, f"interp{k}_f0": f0, f"interp{k}_f": ff})
        meta.append({"kernel": "interpolate_add_to", "id": k, "ofst_i": oi, "ofst_j": oj})
        k += 1

    # split(A_in, A_out, rank)
    k = 0
    for nin, nout in [(17, 10), (33, 18)]:
        A = rnd((nin, nin))
        for rank in range(4):
            o = np.zeros((nout, nout))
            nm.split(A, o, rank)
            out.update({f"split{k}_in": A, f"split{k}_out": o})
            meta.append({"kernel": "split", "id": k, "rank": rank})
            k += 1

    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **out)
    print("kernels.npz:", len(meta), "cases")


# --------------------------------------------------------------------------- solver runs


def test_rhs(n):
    """The eight-vortex right-hand side of test/test.py:40-58."""
    nx0 = 2 ** (n + 1) + 1
    x = np.linspace(-10, 10, nx0)
    y = np.linspace(-10, 10, nx0)
    h = y[1] - y[0]
    X, Y = np.meshgrid(x, y)
    b0 = np.zeros_like(X)
    sign = 1
    for xv, yv in zip([4, -2, 6, 3, 3, 5, -5, -7], [-2, 6, 6, -5, 5, -5, -2, -2]):
        r = (X - xv) ** 2 + (Y - yv) ** 2
        b0 += sign * 50 * np.exp(-r * 7)
        sign *= -1
    return b0, h


def main_rhs(n):
    """The single Gaussian of core/parallel_multigrid.py:359-372."""
    nx0 = 2 ** (n + 1) + 1
    x = np.linspace(-10, 10, nx0)
    y = np.linspace(-10, 10, nx0)
    h = y[1] - y[0]
    X, Y = np.meshgrid(x, y)
    r = X ** 2 + Y ** 2
    return 50 * np.exp(-r * 4), h


def run_solver(n, n_para, rhs, eps, nsolves, mutate):
    """Run the reference solver on 4 in-process ranks; return everything observable."""
    nm, pm, MPI = _load_reference()
    comm = MPI.COMM_WORLD
    b0, h = (test_rhs if rhs == "test" else main_rhs)(n)
    _check_h(h)
    for i in range(n + 2):
        _check_h(h * 2 ** i)
    nx1 = 2 ** n + 1
    res = [dict() for _ in range(4)]
    solvers = [None] * 4
    errs_cycle = [[] for _ in range(4)]  # np.amax(|r_sb[0]|) at each convergence vote
    r_at_restrict = [[] for _ in range(4)]  # max|r_sb[0]| when coarse() is called on it

    def hook(rank):
        mg = solvers[rank]
        errs_cycle[rank].append(float(np.amax(np.abs(mg.r_sb[0]))))

    comm.allreduce_hook = hook
    orig_coarse = pm.coarse

    def coarse_probe(a, a_crs, oi, oj):
        mg = solvers[_tls.rank]
        if mg is not None and a is mg.r_sb[0]:
            r_at_restrict[_tls.rank].append(float(np.max(np.abs(a))))
        return orig_coarse(a, a_crs, oi, oj)

    pm.coarse = coarse_probe
    failures = []

    def worker(rank):
        try:
            _tls.rank = rank
            b = np.zeros((nx1 + 1, nx1 + 1))
            nm.split(b0, b, rank)
            a = np.zeros_like(b)
            R = np.zeros_like(a)
            mg = pm.Multigrid(b, a, R, h, eps, n, n_para)
            solvers[rank] = mg
            its, errs = [], []
            for _ in range(nsolves):
                if mutate:  # test/test.py:77-78
                    b[1:-1] += b[0:-2]
                    b[:] /= 2
                errs.append(float(mg.solve()))
                its.append(int(mg.it))
            res[rank] = {"x": a, "r": R, "b": b, "its": its, "errs": errs, "mg": mg}
        except BaseException as e:  # noqa
            failures.append((rank, repr(e)))
            comm.barrier.abort()
            raise

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    pm.coarse = orig_coarse
    comm.allreduce_hook = None
    assert not failures, failures
    return b0, h, res, errs_cycle, r_at_restrict


def gen_solve(case):
    name, n, n_para, rhs, eps, nsolves, mutate, keep = case
    b0, h, res, errs_cycle, r_at_restrict = run_solver(n, n_para, rhs, eps, nsolves, mutate)
    out = {
        "n": n, "n_para": n_para, "rhs": rhs, "eps": eps, "nsolves": nsolves, "mutate": int(mutate),
        "h": h,
        "its": np.array([res[k]["its"] for k in range(4)]),
        "errs": np.array([res[k]["errs"] for k in range(4)]),
        "err_cycle": np.array(errs_cycle),          # [rank, cycle] across all solves, in order
        "r_at_restrict": np.array(r_at_restrict),   # [rank, cycle]
    }
    mg0 = res[0]["mg"]
    if keep == "blocks":
        for k in range(4):
            out[f"x{k}"] = res[k]["x"]
            out[f"r{k}"] = res[k]["r"]
        # whole-domain level arrays of rank 0 after the last cycle (replicated on all ranks)
        for i, (xw, bw, rw) in enumerate(zip(mg0.x_wl, mg0.b_wl, mg0.r_wl)):
            out[f"x_wl{i}"], out[f"b_wl{i}"], out[f"r_wl{i}"] = xw, bw, rw
    else:  # large case: keep a strided sample of every block
        for k in range(4):
            out[f"x{k}_s8"] = res[k]["x"][::8, ::8].copy()
            out[f"r{k}_s8"] = res[k]["r"][::8, ::8].copy()
            out[f"x{k}_sum"] = np.array(float(np.sum(res[k]["x"])))
            out[f"x{k}_abssum"] = np.array(float(np.sum(np.abs(res[k]["x"]))))
    np.savez_compressed(os.path.join(OUT, f"solve_{name}.npz"), **out)
    return name, out["its"][0].tolist(), out["errs"][:, -1].tolist()


def gen_layout():
    """Shapes and mesh widths of every level (Multigrid.__init__, parallel_multigrid.py:153-230)."""
    nm, pm, MPI = _load_reference()
    _tls.rank = 0
    layout = []
    for n, n_para in [(4, 1), (4, 3), (5, 2), (5, 4), (6, 3), (6, 5), (7, 4), (8, 4), (8, 7), (9, 5)]:
        z = np.zeros((2 ** n + 2, 2 ** n + 2))
        mg = pm.Multigrid(z, z.copy(), z.copy(), 1.0, 0.0, n, n_para)
        layout.append({
            "n": n, "n_para": n_para,
            "x_sb": [list(a.shape) for a in mg.x_sb], "b_sb": [list(a.shape) for a in mg.b_sb],
            "r_sb": [list(a.shape) for a in mg.r_sb], "x_wl": [list(a.shape) for a in mg.x_wl],
            "b_wl": [list(a.shape) for a in mg.b_wl], "h_sb": list(map(float, mg.h_sb)),
            "h_wl": list(map(float, mg.h_wl)),
        })
    offs = []
    for rank in range(4):
        _tls.rank = rank
        z = np.zeros((18, 18))
        mg = pm.Multigrid(z, z.copy(), z.copy(), 1.0, 0.0, 4, 2)
        offs.append({"rank": rank, "i": mg.ofst["i"], "j": mg.ofst["j"],
                     "neighbours": mg.bufs_x[0].neighbours})
    with open(os.path.join(OUT, "layout.json"), "w") as f:
        json.dump({"levels": layout, "offsets": offs}, f, indent=1)
    print("layout.json:", len(layout), "cases")


# --------------------------------------------------------------------------- undecomposed runs


def run_global(b0, n_para, h0, ncycles, x_init=None, fix_coarse_ascent=False):
    """The schedule of Multigrid.solve() (core/parallel_multigrid.py:264-347) driven on ONE undecomposed grid
    with the reference's own kernels (the imported numba_modules functions): offsets 0, no exchange, no
    gather/split -- the restriction of the last Jacobi level lands in b_wl[0] directly and x_wl[0] is the
    correction.  This is the schedule the GPU slab engine runs; the arithmetic is the reference's.
    Level sizes follow __init__ (:182-219) generalised to rectangles: Jacobi levels ((ny-1)>>l)+1, then
    whole-domain levels down to 2 cells per side."""
    nm, _, _ = _load_reference()
    ny0, nx0 = b0.shape
    n1, n2 = 3, 6  # :255-256
    shp = lambda l: (((ny0 - 1) >> l) + 1, ((nx0 - 1) >> l) + 1)
    x = [np.zeros(shp(l)) for l in range(n_para)]
    b = [np.zeros(shp(l)) for l in range(n_para)]
    r = [np.zeros(shp(l)) for l in range(n_para)]
    h = [_check_h(h0 * 2 ** l) for l in range(n_para)]
    b[0][:] = b0
    if x_init is not None:
        x[0][:] = x_init
    xw, bw, rw, hw = [], [], [], []
    l = n_para
    while ((ny0 - 1) >> l) >= 2 and ((nx0 - 1) >> l) >= 2 and ((ny0 - 1) >> l) << l == ny0 - 1 and ((nx0 - 1) >> l) << l == nx0 - 1:
        xw.append(np.zeros(shp(l))), bw.append(np.zeros(shp(l))), rw.append(np.zeros(shp(l)))
        hw.append(_check_h(h0 * 2 ** l))
        l += 1
    L = len(xw)

    def smooth_parallel(bb, xx, hh, rr, iterations):  # :100-114 without the exchanges
        x_temp = np.zeros_like(xx)
        x_temp[:] = xx
        for _ in range(iterations):
            nm.smooth_sweep_jacobi(bb, xx, x_temp, hh)
        nm.residual(rr, xx, bb, hh)

    errs = []
    smooth_parallel(b[0], x[0], h[0], r[0], 1)  # :264-266
    for _ in range(ncycles):
        for lv in range(n_para):  # :272-288
            if lv > 0:
                x[lv][:] = 0
                smooth_parallel(b[lv], x[lv], h[lv], r[lv], n1)
            nm.coarse(r[lv], b[lv + 1] if lv + 1 < n_para else bw[0], 0, 0)
        xw[0][:] = 0  # :291-293
        nm.smooth_altern(bw[0], xw[0], hw[0], rw[0], n1)
        for i in range(1, L):  # :294-297
            xw[i][:] = 0
            nm.coarse(rw[i - 1], bw[i], 0, 0)
            nm.smooth_altern(bw[i], xw[i], hw[i], rw[i], n1)
        for i in range(2, L + (1 if fix_coarse_ascent else 0)):  # :302-304 (range(2, n - n_para + 1) = range(2, L))
            nm.interpolate_add_to(xw[-i + 1], xw[-i], 0, 0)
            nm.smooth_altern(bw[-i], xw[-i], hw[-i], rw[-i], n2)
        for lv in range(n_para - 1, -1, -1):  # :307-333
            nm.interpolate_add_to(xw[0] if lv == n_para - 1 else x[lv + 1], x[lv], 0, 0)
            smooth_parallel(b[lv], x[lv], h[lv], r[lv], n2)
        errs.append(float(np.amax(np.abs(r[0]))))  # :336
    return {"x": x[0], "r": r[0], "errs": np.array(errs), "x_last": x[-1], "b_last": b[-1],
            **{f"xw{i}": a for i, a in enumerate(xw)}, **{f"bw{i}": a for i, a in enumerate(bw)},
            **{f"rw{i}": a for i, a in enumerate(rw)}}


GLOBAL_CASES = [
    # name, ny, nx, n_para, right-hand side, cycles, fix_coarse_ascent
    ("t33_p1", 33, 33, 1, "test", 4, False),
    ("t33_p3", 33, 33, 3, "test", 4, False),
    ("m65_p2", 65, 65, 2, "main", 4, False),
    ("t65_p4", 65, 65, 4, "test", 3, False),
    ("r65x33_p2", 65, 33, 2, "random", 3, False),
    ("r33x129_p3", 33, 129, 3, "random", 3, False),
    ("t129_p3", 129, 129, 3, "test", 3, False),
    ("m129_p3_fix", 129, 129, 3, "main", 3, True),
    ("rx65_p3", 65, 65, 3, "random_x", 3, False),   # non-zero Dirichlet values and initial guess in x
]


def gen_global(case):
    name, ny, nx, n_para, rhs, ncycles, fix = case
    x_init = None
    if rhs in ("test", "main"):
        n = int(np.log2(nx - 1)) - 1
        b0, h = (test_rhs if rhs == "test" else main_rhs)(n)
        assert b0.shape == (ny, nx)
    else:
        rng = np.random.default_rng(len(name) * 7 + ny + nx)
        b0, h = rng.standard_normal((ny, nx)), 0.37
        if rhs == "random_x":
            x_init = rng.standard_normal((ny, nx))
    out = run_global(b0, n_para, h, ncycles, x_init, fix)
    out.update({"b0": b0, "h": h, "n_para": n_para, "ncycles": ncycles, "fix": int(fix)})
    if x_init is not None:
        out["x_init"] = x_init
    np.savez_compressed(os.path.join(OUT, f"global_{name}.npz"), **out)
    return name, out["errs"].tolist()


SOLVE_CASES = [
    # name, n, n_para, rhs, eps, nsolves, mutate(test.py:77-78), keep
    ("t_n4_p1", 4, 1, "test", 0.025, 10, True, "blocks"),   # test.py's ten warm-started solves
    ("t_n4_p3", 4, 3, "test", 0.025, 1, True, "blocks"),
    ("t_n4_p2_raw", 4, 2, "test", 0.025, 1, False, "blocks"),  # un-mutated RHS: under-resolved at n = 4
    ("t_n5_p2", 5, 2, "test", 0.025, 3, True, "blocks"),
    ("t_n5_p4", 5, 4, "test", 0.025, 2, True, "blocks"),
    ("t_n6_p3", 6, 3, "test", 0.025, 2, True, "blocks"),
    ("t_n6_p5", 6, 5, "test", 0.025, 1, True, "blocks"),
    ("m_n5_p2", 5, 2, "main", 0.025, 2, False, "blocks"),
    ("m_n5_p3_cap", 5, 3, "main", 1e-8, 1, False, "blocks"),  # hits the 100-cycle cap
    ("m_n6_p3", 6, 3, "main", 0.025, 1, False, "blocks"),
    ("t_n7_p4", 7, 4, "test", 0.025, 2, True, "blocks"),
    ("t_n8_p4", 8, 4, "test", 0.025, 2, True, "sample"),     # the reference's own test size
    ("m_n8_p4", 8, 4, "main", 0.025, 2, False, "sample"),    # the module's own __main__ (parallel_multigrid.py:351-415)
]


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["kernels", "layout", "global", "solves"]
    if "kernels" in which:
        gen_kernels()
    if "layout" in which:
        gen_layout()
    if "global" in which:
        with ProcessPoolExecutor(max_workers=6) as ex:
            for r in ex.map(gen_global, GLOBAL_CASES):
                print("global", *r, flush=True)
    if "solves" in which:
        names = [a for a in which if a.startswith("solve:")]
        cases = [c for c in SOLVE_CASES if not names or ("solve:" + c[0]) in names]
        with ProcessPoolExecutor(max_workers=6) as ex:
            for r in ex.map(gen_solve, cases):
                print("solve", *r, flush=True)
