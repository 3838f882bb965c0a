"""HIP kernels (through the C ABI) against the golden vectors and against the CPU oracle.

Bit-exact (np.array_equal) everywhere: the kernels keep the reference's expression order and are
built without multiply-add fusion.  Marked gpu: runs on the MI355X box only.
"""
import json
import os

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
K = np.load(os.path.join(GOLDEN, "kernels.npz"))
META = json.loads(str(K["meta"]))


def cases(kernel):
    return [m for m in META if m["kernel"] == kernel]


@pytest.fixture(scope="module")
def mm():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU"
    from multigrid_amd import multigrid_module
    return multigrid_module


@pytest.fixture(scope="module")
def L():
    from multigrid_amd import _lib
    return _lib


def padded(a, ld=None, device="cuda"):
    """Device copy of `a` inside a wider allocation: a view with row stride `ld` (16-element
    multiple by default, which makes the widest vector path eligible)."""
    import torch
    ny, nx = a.shape
    if ld is None:
        ld = (nx + 15) // 16 * 16
    buf = torch.zeros((ny, ld), dtype=torch.float64, device=device)
    buf[:, :nx] = torch.from_numpy(np.ascontiguousarray(a))
    return buf[:, :nx]


# ------------------------------------------------------------------ golden vectors, numpy in/out


@pytest.mark.parametrize("m", cases("residual"), ids=lambda m: f"residual{m['id']}")
def test_golden_residual(mm, m):
    k = m["id"]
    r = K[f"residual{k}_r0"].copy()
    mm.residual(r, K[f"residual{k}_a"], K[f"residual{k}_b"], m["h"])
    assert np.array_equal(r, K[f"residual{k}_r"])


@pytest.mark.parametrize("m", cases("laplacian"), ids=lambda m: f"laplacian{m['id']}")
def test_golden_laplacian(mm, m):
    k = m["id"]
    l = K[f"laplacian{k}_l0"].copy()
    mm.laplacian(K[f"laplacian{k}_a"], l, m["h"])
    assert np.array_equal(l, K[f"laplacian{k}_l"])


@pytest.mark.parametrize("m", cases("smooth_sweep_jacobi"), ids=lambda m: f"jacobi{m['id']}")
def test_golden_jacobi(mm, m):
    k = m["id"]
    a, t = K[f"jacobi{k}_a0"].copy(), K[f"jacobi{k}_t0"].copy()
    mm.smooth_sweep_jacobi(K[f"jacobi{k}_b"], a, t, m["h"])
    assert np.array_equal(a, K[f"jacobi{k}_a"])
    assert np.array_equal(t, K[f"jacobi{k}_t"])


@pytest.mark.parametrize("m", cases("smooth") + cases("smooth_altern"), ids=lambda m: f"{m['kernel']}{m['id']}")
def test_golden_sor(mm, m):
    k = m["id"]
    x, r = K[f"sor{k}_x0"].copy(), K[f"sor{k}_r0"].copy()
    getattr(mm, m["kernel"])(K[f"sor{k}_b"], x, m["h"], r, m["iterations"])
    assert np.array_equal(x, K[f"sor{k}_x"])
    assert np.array_equal(r, K[f"sor{k}_r"])


@pytest.mark.parametrize("m", cases("coarse"), ids=lambda m: f"coarse{m['id']}")
def test_golden_coarse(mm, m):
    k = m["id"]
    c = K[f"coarse{k}_c0"].copy()
    mm.coarse(K[f"coarse{k}_f"], c, m["ofst_i"], m["ofst_j"])
    assert np.array_equal(c, K[f"coarse{k}_c"])


@pytest.mark.parametrize("m", cases("interpolate_add_to"), ids=lambda m: f"interp{m['id']}")
def test_golden_interpolate(mm, m):
    k = m["id"]
    f = K[f"interp{k}_f0"].copy()
    mm.interpolate_add_to(K[f"interp{k}_c"], f, m["ofst_i"], m["ofst_j"])
    assert np.array_equal(f, K[f"interp{k}_f"])


@pytest.mark.parametrize("m", cases("split"), ids=lambda m: f"split{m['id']}")
def test_golden_split(mm, m):
    k = m["id"]
    o = np.zeros_like(K[f"split{k}_out"])
    mm.split(K[f"split{k}_in"], o, m["rank"])
    assert np.array_equal(o, K[f"split{k}_out"])


# ------------------------------------------------------------------ oracle, device tensors, all vector widths

SHAPES = [(3, 3), (5, 9), (33, 33), (130, 130), (257, 513), (514, 1030), (100, 2051), (40, 4099)]


@pytest.mark.parametrize("shape", SHAPES, ids=str)
@pytest.mark.parametrize("vec,unroll", [(1, 1), (1, 2), (2, 1), (2, 2), (2, 4)])
def test_stencils_vs_oracle(mm, L, shape, vec, unroll):
    rng = np.random.default_rng(hash(shape) % 1000 + vec * 8 + unroll)
    a, b = rng.standard_normal(shape), rng.standard_normal(shape)
    r0, t0 = rng.standard_normal(shape), None
    h = 0.3
    L.check(L.lib().mg_set_tuning(b"vector", vec))
    L.check(L.lib().mg_set_tuning(b"unroll", unroll))
    try:
        # residual
        r_ref = r0.copy()
        oracle.residual(r_ref, a, b, h)
        da, db, dr = padded(a), padded(b), padded(r0)
        mm.residual(dr, da, db, h)
        assert np.array_equal(dr.cpu().numpy(), r_ref)
        # jacobi (module semantics: a_temp pre-filled with a)
        a_ref, t_ref = a.copy(), a.copy()
        oracle.smooth_sweep_jacobi(b, a_ref, t_ref, h)
        da, dt = padded(a), padded(a)
        mm.smooth_sweep_jacobi(db, da, dt, h)
        assert np.array_equal(da.cpu().numpy(), a_ref)
        assert np.array_equal(dt.cpu().numpy(), t_ref)
        # laplacian
        l_ref = r0.copy()
        oracle.laplacian(a, l_ref, h)
        dl = padded(r0)
        mm.laplacian(padded(a), dl, h)
        assert np.array_equal(dl.cpu().numpy(), l_ref)
    finally:
        L.check(L.lib().mg_set_tuning(b"vector", 2))
        L.check(L.lib().mg_set_tuning(b"unroll", 2))


@pytest.mark.parametrize("om", [1.0, 0.8])
def test_jacobi_row_range_and_from_zero(L, om):
    """Row-range launches touch only their rows; the from-zero sweep equals a sweep on x = 0.  om < 1: the
    damped sweep x + om * (t - x) against the oracle's."""
    import torch
    rng = np.random.default_rng(5)
    ny, nx = 70, 300
    a, b = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    h = 0.7
    ref, t = a.copy(), a.copy()
    oracle.smooth_sweep_jacobi(b, ref, t, h, om)
    da, db = padded(a), padded(b)
    out = padded(np.full((ny, nx), 7.0))
    ld = out.stride(0)
    st = torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    L.check(lib.mg_jacobi(db.data_ptr(), da.data_ptr(), out.data_ptr(), ny, nx, ld, 10, 33, h, om, 1, 0, 0, st))
    o = out.cpu().numpy()
    assert np.array_equal(o[10:33, 1:-1], ref[10:33, 1:-1])
    o[10:33, 1:-1] = 7.0
    assert np.all(o == 7.0)
    # two launches covering disjoint row ranges == one full sweep
    L.check(lib.mg_jacobi(db.data_ptr(), da.data_ptr(), out.data_ptr(), ny, nx, ld, 1, 10, h, om, 1, 0, 0, st))
    L.check(lib.mg_jacobi(db.data_ptr(), da.data_ptr(), out.data_ptr(), ny, nx, ld, 10, ny - 1, h, om, 1, 0, 0, st))
    assert np.array_equal(out.cpu().numpy()[1:-1, 1:-1], ref[1:-1, 1:-1])
    # from zero
    z, tz = np.zeros((ny, nx)), np.zeros((ny, nx))
    oracle.smooth_sweep_jacobi(b, z, tz, h, om)
    out2 = padded(np.zeros((ny, nx)))
    L.check(lib.mg_jacobi_from_zero(db.data_ptr(), out2.data_ptr(), ny, nx, ld, 1, ny - 1, h, om, 1, 0, 0, st))
    got = out2.cpu().numpy()
    assert np.array_equal(got, z)
    assert np.array_equal(np.signbit(got), np.signbit(z))


def test_residual_norm(L):
    import ctypes as C
    import torch
    rng = np.random.default_rng(9)
    ny, nx = 515, 1027
    a, b = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    h = 0.25
    r_ref = np.zeros((ny, nx))
    oracle.residual(r_ref, a, b, h)
    da, db, dr = padded(a), padded(b), padded(np.zeros((ny, nx)))
    slots = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    lib = L.lib()
    out = C.c_double()
    for flags in (0, L.FLAG_FINEST):
        L.check(lib.mg_norm_reset(slots.data_ptr(), st))
        L.check(lib.mg_residual(dr.data_ptr(), da.data_ptr(), db.data_ptr(), ny, nx, dr.stride(0), 1, ny - 1, h,
                                slots.data_ptr(), 1, 0, flags, st))
        L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(out), st))
        assert out.value == np.abs(r_ref).max()
        assert np.array_equal(dr.cpu().numpy(), r_ref)
    # whole-array max, boundary included
    big = a.copy()
    big[0, 5] = -1e9
    L.check(lib.mg_norm_reset(slots.data_ptr(), st))
    dbig = padded(big)
    L.check(lib.mg_maxabs(dbig.data_ptr(), ny, nx, dbig.stride(0), slots.data_ptr(), st))
    L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(out), st))
    assert out.value == 1e9
    # a NaN wins, as in np.amax
    a[100, 200] = np.nan
    da = padded(a)
    L.check(lib.mg_norm_reset(slots.data_ptr(), st))
    L.check(lib.mg_residual(dr.data_ptr(), da.data_ptr(), db.data_ptr(), ny, nx, dr.stride(0), 1, ny - 1, h,
                            slots.data_ptr(), 1, 0, 0, st))
    L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(out), st))
    assert np.isnan(out.value)


TRANSFER = [((18, 18), (10, 10)), ((34, 66), (18, 34)), ((130, 130), (66, 66)), ((258, 514), (130, 258)),
            ((1026, 1026), (514, 514))]


@pytest.mark.parametrize("fshape,cshape", TRANSFER, ids=str)
@pytest.mark.parametrize("oi", [0, -1])
@pytest.mark.parametrize("oj", [0, -1])
@pytest.mark.parametrize("pad", [True, False])
def test_transfer_vs_oracle(mm, fshape, cshape, oi, oj, pad):
    rng = np.random.default_rng(abs(hash((fshape, oi, oj))) % 10000)
    f, c0 = rng.standard_normal(fshape), rng.standard_normal(cshape)
    c_ref = c0.copy()
    oracle.coarse(f, c_ref, oi, oj)
    mk = (lambda a: padded(a)) if pad else (lambda a: padded(a, ld=a.shape[1] + 1))
    dc = mk(c0)
    mm.coarse(mk(f), dc, oi, oj)
    assert np.array_equal(dc.cpu().numpy(), c_ref)
    cc, f0 = rng.standard_normal(cshape), rng.standard_normal(fshape)
    f_ref = f0.copy()
    oracle.interpolate_add_to(cc, f_ref, oi, oj)
    df = mk(f0)
    mm.interpolate_add_to(mk(cc), df, oi, oj)
    assert np.array_equal(df.cpu().numpy(), f_ref)


@pytest.mark.parametrize("fshape,cshape", [((17, 17), (9, 9)), ((5, 5), (3, 3)), ((513, 1025), (257, 513)),
                                           ((2049, 2049), (1025, 1025))], ids=str)
def test_transfer_whole_domain(mm, fshape, cshape):
    """Offsets 0,0 on grids that include their boundary (the whole-domain levels)."""
    rng = np.random.default_rng(3)
    f, c0 = rng.standard_normal(fshape), rng.standard_normal(cshape)
    c_ref = c0.copy()
    oracle.coarse(f, c_ref, 0, 0)
    dc = padded(c0)
    mm.coarse(padded(f), dc, 0, 0)
    assert np.array_equal(dc.cpu().numpy(), c_ref)
    f_ref = f.copy()
    oracle.interpolate_add_to(c0, f_ref, 0, 0)
    df = padded(f)
    mm.interpolate_add_to(padded(c0), df, 0, 0)
    assert np.array_equal(df.cpu().numpy(), f_ref)


SOR_SHAPES = [(3, 3), (5, 5), (4, 7), (9, 9), (33, 33), (65, 257), (129, 129), (141, 139), (1100, 9), (257, 257), (1030, 40),
              (65, 65), (66, 40), (64, 200), (5, 1000), (128, 128), (130, 90), (200, 90), (257, 65), (258, 20), (17, 3), (3, 17),
              (129, 17), (129, 33), (65, 9), (131, 4), (1025, 9)]


@pytest.mark.parametrize("shape", SOR_SHAPES, ids=str)
@pytest.mark.parametrize("its", [1, 3])
@pytest.mark.parametrize("single_wave,transpose", [(2, 1), (2, 0), (0, 1)])
def test_sor_lexicographic_vs_oracle(mm, L, shape, its, single_wave, transpose):
    """Wavefront execution (one wave with lane shifts -- along the rows, or along the columns of a grid that has
    more rows than columns --, one thread per row in LDS, global memory) == row-major Gauss-Seidel, bit for bit."""
    if transpose == 0 and shape[0] <= shape[1]:
        pytest.skip("only grids with more rows than columns are walked transposed")
    rng = np.random.default_rng(shape[0] * 7 + its)
    b, x0, r0 = rng.standard_normal(shape), rng.standard_normal(shape), rng.standard_normal(shape)
    L.check(L.lib().mg_set_tuning(b"sor_single_wave", single_wave))
    L.check(L.lib().mg_set_tuning(b"sor_transpose", transpose))
    try:
        for fn in ("smooth", "smooth_altern"):
            x_ref, r_ref = x0.copy(), r0.copy()
            getattr(oracle, fn)(b, x_ref, 0.3, r_ref, its)
            dx, dr = padded(x0), padded(r0)
            getattr(mm, fn)(padded(b), dx, 0.3, dr, its)
            assert np.array_equal(dx.cpu().numpy(), x_ref), fn
            assert np.array_equal(dr.cpu().numpy(), r_ref), fn
    finally:
        L.check(L.lib().mg_set_tuning(b"sor_single_wave", 1))
        L.check(L.lib().mg_set_tuning(b"sor_transpose", 1))


@pytest.mark.parametrize("shape", [(3, 3), (9, 9), (33, 65), (129, 129), (513, 513)], ids=str)
def test_sor_redblack_vs_oracle(mm, shape):
    rng = np.random.default_rng(11)
    b, x0, r0 = rng.standard_normal(shape), rng.standard_normal(shape), rng.standard_normal(shape)
    x_ref, r_ref = x0.copy(), r0.copy()
    oracle.smooth_altern_rb(b, x_ref, 0.3, r_ref, 3)
    dx, dr = padded(x0), padded(r0)
    mm.smooth_altern(padded(b), dx, 0.3, dr, 3, ordering="redblack")
    assert np.array_equal(dx.cpu().numpy(), x_ref)
    assert np.array_equal(dr.cpu().numpy(), r_ref)


def test_batched_kernels(L):
    om = 1.0
    """nbatch > 1: every grid of the batch gets the single-grid result."""
    import torch
    rng = np.random.default_rng(21)
    nb, ny, nx = 5, 33, 33
    ld = 48
    stride = ny * ld
    h = 0.5

    def dev(a3):
        buf = torch.zeros((nb, ny, ld), dtype=torch.float64, device="cuda")
        buf[:, :, :nx] = torch.from_numpy(a3)
        return buf

    b, x = rng.standard_normal((nb, ny, nx)), rng.standard_normal((nb, ny, nx))
    db, dx, do = dev(b), dev(x), dev(x)
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mg_jacobi(db.data_ptr(), dx.data_ptr(), do.data_ptr(), ny, nx, ld, 1, ny - 1, h, om, nb, stride, 0, st))
    got = do.cpu().numpy()[:, :, :nx]
    for k in range(nb):
        ref, t = x[k].copy(), x[k].copy()
        oracle.smooth_sweep_jacobi(b[k], ref, t, h)
        assert np.array_equal(got[k], ref)
    dr = dev(np.zeros((nb, ny, nx)))
    L.check(lib.mg_sor(db.data_ptr(), dx.data_ptr(), dr.data_ptr(), ny, nx, ld, h, 3, 1, L.SOR_LEXICOGRAPHIC, nb, stride,
                       st))
    gx, gr = dx.cpu().numpy()[:, :, :nx], dr.cpu().numpy()[:, :, :nx]
    for k in range(nb):
        xr, rr = x[k].copy(), np.zeros((ny, nx))
        oracle.smooth_altern(b[k], xr, h, rr, 3)
        assert np.array_equal(gx[k], xr)
        assert np.array_equal(gr[k], rr)
    # restriction / prolongation, batched
    cny = cnx = 17
    cld = 32
    c = dev(np.zeros((nb, ny, nx)))[:, :cny, :cld].contiguous()
    L.check(lib.mg_restrict(dx.data_ptr(), ny, nx, ld, c.data_ptr(), cny, cnx, cld, 1, cny - 1, 0, 0, nb, stride,
                            cny * cld, st))
    gc = c.cpu().numpy()[:, :, :cnx]
    for k in range(nb):
        cr = np.zeros((cny, cnx))
        oracle.coarse(gx[k].copy(), cr, 0, 0)
        assert np.array_equal(gc[k], cr)
    L.check(lib.mg_prolong_add(c.data_ptr(), cny, cnx, cld, dx.data_ptr(), ny, nx, ld, 0, ny, 0, nx - 1, 0, 0, nb,
                               cny * cld, stride, st))
    gx2 = dx.cpu().numpy()[:, :, :nx]
    for k in range(nb):
        fr = gx[k].copy()
        oracle.interpolate_add_to(np.ascontiguousarray(gc[k]), fr, 0, 0)
        assert np.array_equal(gx2[k], fr)


def test_error_behaviour(mm, L):
    om = 1.0
    """Wrong dtype/rank -> TypeError (as the compiled reference module); bad shapes are rejected
    on the host before any launch."""
    import torch
    a = np.zeros((10, 10))
    with pytest.raises(TypeError):
        mm.residual(a.astype(np.float32), a, a, 1.0)
    with pytest.raises(TypeError):
        mm.residual(a[0], a, a, 1.0)
    with pytest.raises(TypeError):
        mm.coarse(torch.zeros(10, 10, dtype=torch.float64), a, 0, 0)  # CPU tensor
    with pytest.raises(L.MgError):
        mm.residual(a, a, np.zeros((10, 11)), 1.0)
    with pytest.raises(L.MgError):
        mm.coarse(np.zeros((10, 10)), np.zeros((10, 10)), 0, 0)  # fine grid too small
    with pytest.raises(L.MgError):
        mm.interpolate_add_to(np.zeros((10, 10)), np.zeros((10, 10)), 0, 0)
    with pytest.raises(L.MgError):
        mm.coarse(np.zeros((18, 18)), np.zeros((10, 10)), 1, 0)
    lib = L.lib()
    d = padded(a)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.mg_jacobi(d.data_ptr(), d.data_ptr(), d.data_ptr(), 10, 10, 16, 1, 9, 1.0, om, 1, 0, 0, st) != 0
    assert b"alias" in lib.mg_last_error()
    assert lib.mg_jacobi(d.data_ptr(), d.data_ptr(), None, 10, 10, 16, 0, 9, 1.0, om, 1, 0, 0, st) != 0
    d2 = torch.zeros((10, 16), dtype=torch.float64, device="cuda")
    for bad in (0.0, 1.5, -0.8, float("nan")):  # the damping factor must lie in (0, 1]
        assert lib.mg_jacobi(d.data_ptr(), d.data_ptr(), d2.data_ptr(), 10, 10, 16, 1, 9, 1.0, bad, 1, 0, 0, st) != 0
        assert lib.mg_jacobi_from_zero(d.data_ptr(), d2.data_ptr(), 10, 10, 16, 1, 9, 1.0, bad, 1, 0, 0, st) != 0
        assert lib.mg_smooth_fused(d.data_ptr(), d.data_ptr(), d2.data_ptr(), None, None, 10, 10, 16, 0, 0, 0, 1, 9, 0, 9, 0, 3, 0,
                                   1.0, bad, None, 1, 0, 0, st) != 0
    # damped passes exist for 1, 3 and 6 sweeps
    assert lib.mg_smooth_fused(d.data_ptr(), d.data_ptr(), d2.data_ptr(), None, None, 10, 10, 16, 0, 0, 0, 1, 9, 0, 9, 0, 2, 0,
                               1.0, 0.8, None, 1, 0, 0, st) != 0
    assert lib.mg_smooth_fused(d.data_ptr(), d.data_ptr(), d2.data_ptr(), None, None, 10, 10, 16, 0, 0, 0, 1, 9, 0, 9, 0, 3, 0,
                               1.0, 0.8, None, 1, 0, 0, st) == 0
    # rank outside 0..3: the reference copies nothing
    o = np.full((5, 5), 3.0)
    mm.split(a, o, 7)
    assert np.all(o == 3.0)


# ------------------------------------------------------------------ fused multi-sweep passes

FUSED_SHAPES = [(9, 9), (40, 121), (70, 249), (66, 250), (300, 257), (130, 1000), (600, 2051), (35, 4099)]


@pytest.mark.parametrize("shape", FUSED_SHAPES, ids=str)
@pytest.mark.parametrize("sweeps", [1, 2, 3, 4, 6])
@pytest.mark.parametrize("mode", ["plain", "zero", "prolong"])
@pytest.mark.parametrize("resid", [False, True])
@pytest.mark.parametrize("unroll", [1, 2])
@pytest.mark.parametrize("om", [1.0, 0.8])
def test_fused_pass_equals_separate_kernels(L, shape, sweeps, mode, resid, unroll, om):
    """mg_smooth_fused == mg_prolong_add, `sweeps` x mg_jacobi, mg_residual, bit for bit (and hence
    == the oracle, which the separate kernels are checked against above)."""
    import ctypes as C
    import torch
    if om != 1.0 and sweeps not in (1, 3, 6):
        pytest.skip("damped passes are built for the depths of the V-cycle")
    ny, nx = shape
    rng = np.random.default_rng(ny * 131 + nx + sweeps)
    h = 0.3
    x = rng.standard_normal(shape)
    b = rng.standard_normal(shape)
    r0 = rng.standard_normal(shape)
    if mode == "zero":
        x[:] = 0
    # slab convention: row 0 is a dummy, row 1 and row ny-1 are Dirichlet rows, fine row t <-> coarse (t+1)/2
    flo, fhi, j0, j1 = 1, ny - 1, 2, ny - 1
    R = ny - 2
    cny, cnx = (R // 2 + 2 if R % 2 == 0 else 0), (nx - 1) // 2 + 1
    prolong = mode == "prolong" and R % 2 == 0 and nx % 2 == 1
    if mode == "prolong" and not prolong:
        pytest.skip("no coarse grid for this shape")
    c = rng.standard_normal((cny, cnx)) if prolong else None
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.mg_set_fused_tuning(unroll, 0))
    db, dx = padded(b), padded(x)
    ld = dx.stride(0)
    dc = padded(c) if prolong else None
    # reference sequence with the separate kernels
    cur = padded(x)
    if prolong:
        L.check(lib.mg_prolong_add(dc.data_ptr(), cny, cnx, dc.stride(0), cur.data_ptr(), ny, nx, ld, j0, j1, 1, nx - 2,
                                   0, -1, 1, 0, 0, st))
    oth = padded(cur.cpu().numpy())
    for _ in range(sweeps):
        L.check(lib.mg_jacobi(db.data_ptr(), cur.data_ptr(), oth.data_ptr(), ny, nx, ld, j0, j1, h, om, 1, 0, 0, st))
        cur, oth = oth, cur
    slots = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    r_ref = padded(r0)
    nrm_ref = C.c_double()
    if resid:
        L.check(lib.mg_residual(r_ref.data_ptr(), cur.data_ptr(), db.data_ptr(), ny, nx, ld, j0, j1, h, slots.data_ptr(),
                                1, 0, 0, st))
        L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(nrm_ref), st))
    # fused
    out = padded(x)  # boundary/fixed cells of the output array must already hold x's values
    rr = padded(r0)
    L.check(lib.mg_norm_reset(slots.data_ptr(), st))
    L.check(lib.mg_smooth_fused(db.data_ptr(), None if mode == "zero" else dx.data_ptr(), out.data_ptr(),
                                rr.data_ptr() if resid else None, dc.data_ptr() if prolong else None, ny, nx, ld,
                                cny if prolong else 0, cnx if prolong else 0, dc.stride(0) if prolong else 0, j0, j1,
                                flo, fhi, -1, sweeps, 0, h, om, slots.data_ptr() if resid else None, 1, 0, 0, st))
    assert np.array_equal(out.cpu().numpy(), cur.cpu().numpy())
    if resid:
        nrm = C.c_double()
        L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(nrm), st))
        assert np.array_equal(rr.cpu().numpy(), r_ref.cpu().numpy())
        assert nrm.value == nrm_ref.value
    L.check(lib.mg_set_fused_tuning(2, 0))


def test_fused_pass_partial_rows(L):
    om = 1.0
    """Producing only rows [j0, j1) needs valid input on D more rows each side and leaves the rest alone."""
    import torch
    rng = np.random.default_rng(77)
    ny, nx, h = 200, 515, 0.4
    x, b = rng.standard_normal((ny, nx)), rng.standard_normal((ny, nx))
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dx = padded(b), padded(x)
    ld = dx.stride(0)
    cur, oth = padded(x), padded(x)
    for _ in range(3):
        L.check(lib.mg_jacobi(db.data_ptr(), cur.data_ptr(), oth.data_ptr(), ny, nx, ld, 1, ny - 1, h, om, 1, 0, 0, st))
        cur, oth = oth, cur
    out = padded(np.full((ny, nx), 5.0))
    L.check(lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), out.data_ptr(), None, None, ny, nx, ld, 0, 0, 0, 60, 131, 0,
                                ny - 1, 0, 3, 0, h, om, None, 1, 0, 0, st))
    got, ref = out.cpu().numpy(), cur.cpu().numpy()
    assert np.array_equal(got[60:131], ref[60:131])
    got[60:131] = 5.0
    assert np.all(got == 5.0)


@pytest.mark.parametrize("mag", [2.0 ** -900, 2.0 ** -40, 2.0 ** 700, 2.0 ** 960])
@pytest.mark.parametrize("sweeps,resid", [(6, False), (6, True), (3, True)])
def test_fused_pass_is_exact_over_the_exponent_range(L, mag, sweeps, resid):
    """The fused pass keeps stage s's values times 4^s (one fma per point instead of multiply, subtract,
    multiply).  That is the reference's arithmetic bit for bit as long as no intermediate leaves the normal
    range: data 2^-900 .. 2^960 times a standard normal sample, where the sequence of single sweeps is the
    check (the mesh width is chosen so that h*h*b stays normal too)."""
    import torch
    rng = np.random.default_rng(123)
    ny, nx = 150, 300
    x, b = rng.standard_normal((ny, nx)) * mag, rng.standard_normal((ny, nx)) * mag
    h = 0.75
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dx = padded(b), padded(x)
    ld = dx.stride(0)
    cur, oth = padded(x), padded(x)
    for _ in range(sweeps):
        L.check(lib.mg_jacobi(db.data_ptr(), cur.data_ptr(), oth.data_ptr(), ny, nx, ld, 1, ny - 1, h, 1.0, 1, 0, 0, st))
        cur, oth = oth, cur
    r_ref = padded(np.zeros((ny, nx)))
    L.check(lib.mg_residual(r_ref.data_ptr(), cur.data_ptr(), db.data_ptr(), ny, nx, ld, 1, ny - 1, h, None, 1, 0, 0, st))
    out, rr = padded(x), padded(np.zeros((ny, nx)))
    L.check(lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), out.data_ptr(), rr.data_ptr() if resid else None, None, ny, nx, ld,
                                0, 0, 0, 1, ny - 1, 0, ny - 1, 0, sweeps, 0, h, 1.0, None, 1, 0, 0, st))
    got = out.cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got).max() > 0
    assert np.array_equal(got, cur.cpu().numpy())
    if resid:
        assert np.array_equal(rr.cpu().numpy(), r_ref.cpu().numpy())


@pytest.mark.parametrize("shape", [(40, 121), (70, 249), (300, 257), (130, 1001), (600, 2051)], ids=str)
@pytest.mark.parametrize("sweeps", [1, 3, 6])
@pytest.mark.parametrize("mode", ["plain", "zero", "prolong"])
@pytest.mark.parametrize("store_r,waves", [(True, 4), (False, 4), (False, 2), (True, 1)])
@pytest.mark.parametrize("om", [1.0, 0.8])
def test_fused_restriction_equals_separate_kernels(L, shape, sweeps, mode, store_r, waves, om):
    """mg_smooth_fused_restrict == mg_smooth_fused followed by mg_restrict, bit for bit, whether or
    not the residual itself is stored, for every number of column strips per workgroup."""
    import ctypes as C
    import torch
    L.check(L.lib().mg_set_tuning(b"fused_waves", waves))
    try:
        _fused_restriction_case(L, C, torch, shape, sweeps, mode, store_r, om)
    finally:
        L.check(L.lib().mg_set_tuning(b"fused_waves", 4))


def _fused_restriction_case(L, C, torch, shape, sweeps, mode, store_r, om):
    ny, nx = shape
    rng = np.random.default_rng(ny + 7 * nx + sweeps)
    h = 0.3
    x = rng.standard_normal(shape)
    b = rng.standard_normal(shape)
    if mode == "zero":
        x[:] = 0
    flo, fhi, j0, j1 = 1, ny - 1, 2, ny - 1
    R = ny - 2
    assert R % 2 == 0 and nx % 2 == 1
    cny, cnx = R // 2 + 2, (nx - 1) // 2 + 1
    c = rng.standard_normal((cny, cnx)) if mode == "prolong" else None
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dx = padded(b), padded(x)
    ld = dx.stride(0)
    dc = padded(c) if c is not None else None
    z = np.zeros(shape)
    zc = np.zeros((cny, cnx))
    args = (db.data_ptr(), None if mode == "zero" else dx.data_ptr())
    cargs = (dc.data_ptr() if c is not None else None, ny, nx, ld, cny if c is not None else 0,
             cnx if c is not None else 0, dc.stride(0) if c is not None else 0, j0, j1, flo, fhi, -1, sweeps)
    # separate: fused smoothing + residual, then the restriction kernel
    o1, r1, cb1 = padded(x), padded(z), padded(zc)
    s1 = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    L.check(lib.mg_smooth_fused(*args, o1.data_ptr(), r1.data_ptr(), *cargs, 0, h, om, s1.data_ptr(), 1, 0, 0, st))
    L.check(lib.mg_restrict(r1.data_ptr(), ny, nx, ld, cb1.data_ptr(), cny, cnx, cb1.stride(0), 2, cny - 1, 0, -1, 1, 0, 0,
                            st))
    # fused
    o2, r2, cb2 = padded(x), padded(z), padded(zc)
    s2 = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    L.check(lib.mg_smooth_fused_restrict(*args, o2.data_ptr(), r2.data_ptr() if store_r else None, *cargs, h, om, s2.data_ptr(),
                                         0, cb2.data_ptr(), cny, cnx, cb2.stride(0), 2, cny - 1, 1, 0, 0, 0, st))
    assert np.array_equal(o2.cpu().numpy(), o1.cpu().numpy())
    assert np.array_equal(cb2.cpu().numpy(), cb1.cpu().numpy())
    n1, n2 = C.c_double(), C.c_double()
    L.check(lib.mg_norm_read(s1.data_ptr(), C.byref(n1), st))
    L.check(lib.mg_norm_read(s2.data_ptr(), C.byref(n2), st))
    assert n1.value == n2.value
    if store_r:
        assert np.array_equal(r2.cpu().numpy(), r1.cpu().numpy())
    else:
        assert not r2.cpu().numpy().any()


@pytest.mark.parametrize("shape", [(10, 9), (18, 17), (34, 33), (40, 121), (70, 249), (300, 257), (130, 1001), (600, 2051),
                                   (258, 4099)], ids=str)
@pytest.mark.parametrize("resid", [False, True])
@pytest.mark.parametrize("om", [1.0, 0.8])
def test_fused_presmoothing_recomputed(L, shape, resid, om):
    """prolong_after = 3, sweeps = 9: 3 sweeps from x = 0, += P(c), 6 sweeps [, residual] in one pass ==
    the separate kernels; and the descent pass that stores no x produces the same coarse right-hand
    side and norm as the one that does."""
    import ctypes as C
    import torch
    ny, nx = shape
    rng = np.random.default_rng(ny + 3 * nx)
    h = 0.3
    b = rng.standard_normal(shape)
    flo, fhi, j0, j1 = 1, ny - 1, 2, ny - 1
    R = ny - 2
    assert R % 2 == 0 and nx % 2 == 1
    cny, cnx = R // 2 + 2, (nx - 1) // 2 + 1
    c = rng.standard_normal((cny, cnx))
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dc = padded(b), padded(c)
    ld = db.stride(0)
    z = np.zeros(shape)
    cur, oth = padded(z), padded(z)
    for k in range(9):
        if k == 3:
            L.check(lib.mg_prolong_add(dc.data_ptr(), cny, cnx, dc.stride(0), cur.data_ptr(), ny, nx, ld, j0, j1, 1, nx - 2,
                                       0, -1, 1, 0, 0, st))
            oth.copy_(cur)
        L.check(lib.mg_jacobi(db.data_ptr(), cur.data_ptr(), oth.data_ptr(), ny, nx, ld, j0, j1, h, om, 1, 0, 0, st))
        cur, oth = oth, cur
    r_ref = padded(z)
    s1 = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    if resid:
        L.check(lib.mg_residual(r_ref.data_ptr(), cur.data_ptr(), db.data_ptr(), ny, nx, ld, j0, j1, h, s1.data_ptr(), 1, 0, 0,
                                st))
    out, rr = padded(z), padded(z)
    s2 = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    L.check(lib.mg_smooth_fused(db.data_ptr(), None, out.data_ptr(), rr.data_ptr() if resid else None, dc.data_ptr(), ny, nx,
                                ld, cny, cnx, dc.stride(0), j0, j1, flo, fhi, -1, 9, 3, h, om, s2.data_ptr() if resid else None,
                                1, 0, 0, st))
    assert np.array_equal(out.cpu().numpy(), cur.cpu().numpy())
    if resid:
        assert np.array_equal(rr.cpu().numpy(), r_ref.cpu().numpy())
        n1, n2 = C.c_double(), C.c_double()
        L.check(lib.mg_norm_read(s1.data_ptr(), C.byref(n1), st))
        L.check(lib.mg_norm_read(s2.data_ptr(), C.byref(n2), st))
        assert n1.value == n2.value
    # descent pass with and without the x output
    zc = np.zeros((cny, cnx))
    o1, cb1, cb2 = padded(z), padded(zc), padded(zc)
    for xo, cb in ((o1, cb1), (None, cb2)):
        L.check(lib.mg_smooth_fused_restrict(db.data_ptr(), None, xo.data_ptr() if xo is not None else None, None, None, ny,
                                             nx, ld, 0, 0, 0, j0, j1, flo, fhi, -1, 3, h, om, None, 0, cb.data_ptr(), cny, cnx,
                                             cb.stride(0), 2, cny - 1, 1, 0, 0, 0, st))
    assert np.array_equal(cb1.cpu().numpy(), cb2.cpu().numpy())
    assert cb1.cpu().numpy().any()
    # rejected combinations
    assert lib.mg_smooth_fused(db.data_ptr(), None, out.data_ptr(), None, dc.data_ptr(), ny, nx, ld, cny, cnx, dc.stride(0),
                               j0, j1, flo, fhi, -1, 6, 3, h, om, None, 1, 0, 0, st) != 0
    assert lib.mg_smooth_fused(db.data_ptr(), out.data_ptr(), rr.data_ptr(), None, dc.data_ptr(), ny, nx, ld, cny, cnx,
                               dc.stride(0), j0, j1, flo, fhi, -1, 9, 3, h, om, None, 1, 0, 0, st) != 0


@pytest.mark.parametrize("shape", [(3, 3), (5, 9), (33, 33), (130, 130), (257, 513), (100, 2051)], ids=str)
def test_gauss_seidel_redblack_vs_oracle(shape):
    """mg_gauss_seidel_rb (colour 0, then colour 1) == the oracle's red-black sweep, bit for bit; a row range leaves
    the other rows alone."""
    import torch
    from multigrid_amd import _lib as L
    lib = L.lib()
    ny, nx = shape
    rng = np.random.default_rng(3)
    b, x = rng.standard_normal(shape), rng.standard_normal(shape)
    want = x.copy()
    oracle.smooth_sweep_rbgs(b, want, 0.3)
    ld = (nx + 15) // 16 * 16
    db = torch.zeros(ny, ld, dtype=torch.float64, device="cuda")
    dx = torch.zeros(ny, ld, dtype=torch.float64, device="cuda")
    db[:, :nx], dx[:, :nx] = torch.from_numpy(b).cuda(), torch.from_numpy(x).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for colour in (0, 1):
        L.check(lib.mg_gauss_seidel_rb(db.data_ptr(), dx.data_ptr(), ny, nx, ld, 1, ny - 1, 0.3, colour, 1, 0, st))
    assert np.array_equal(dx[:, :nx].cpu().numpy(), want)
    if ny >= 33:
        dx[:, :nx] = torch.from_numpy(x).cuda()
        L.check(lib.mg_gauss_seidel_rb(db.data_ptr(), dx.data_ptr(), ny, nx, ld, 10, 20, 0.3, 0, 1, 0, st))
        got = dx[:, :nx].cpu().numpy()
        assert np.array_equal(got[:10], x[:10]) and np.array_equal(got[20:], x[20:]) and not np.array_equal(got[10:20], x[10:20])
    assert lib.mg_gauss_seidel_rb(db.data_ptr(), dx.data_ptr(), ny, nx, ld, 0, ny, 0.3, 0, 1, 0, st) != 0


# the modes of the fused pass that the V-cycle launches, each against the oracle's kernels directly:
# (name, sweeps, prolong_after, start from zero, prolongation, residual, restriction, x stored)
FUSED_MODES = [("finest", 6, 0, False, True, True, True, True), ("opening", 1, 0, False, False, True, True, True),
               ("descent", 3, 0, True, False, True, True, True), ("descent without x", 3, 0, True, False, True, True, False),
               ("ascent", 6, 0, False, True, False, False, True), ("ascent recomputed", 9, 3, True, True, False, False, True),
               ("plain", 3, 0, False, False, False, False, True)]


@pytest.mark.parametrize("shape", [(33, 33), (129, 257), (257, 1025), (65, 4097)], ids=str)
@pytest.mark.parametrize("mode", FUSED_MODES, ids=lambda m: m[0])
def test_fused_pass_equals_oracle_kernels(L, shape, mode):
    """Every mode of mg_smooth_fused / mg_smooth_fused_restrict that the cycle uses == the same sequence of the
    oracle's kernels (interpolate_add_to, smooth_sweep_jacobi, residual, coarse: core/numba_modules.py:53-66, :35-50,
    :132-160, :163-180) on the whole grid, bit for bit: x, r, the max-norm and the coarse right-hand side."""
    _fused_vs_oracle(L, shape, mode)


# tuning knobs change how a pass is cut into tiles, how its waves synchronise and which kind of store it uses,
# never a value: (key, value, key's default)
KNOBS = [("stream_min_mib", 0, 256), ("fused_lockstep", 0, 1), ("fused_lockstep", 3, 1), ("fused_rows", 7, 0), ("fused_rows", 24, 0),
         ("fused_rows", 256, 0), ("fused_rows_shallow", 64, 32), ("fused_rows_deep", 32, 128), ("fused_waves", 1, 4),
         ("fused_waves", 2, 4), ("xcd_stagger", -1, 0), ("xcd_stagger", 3, 0), ("fused_min_rows", 4, 8), ("fused_min_rows", 2, 8), ("fused_min_rows", 1, 8),
         ("fused_min_blocks", 1, 2048)]


@pytest.mark.parametrize("knob", KNOBS, ids=lambda k: f"{k[0]}={k[1]}")
@pytest.mark.parametrize("mode", FUSED_MODES, ids=lambda m: m[0])
def test_fused_pass_equals_oracle_under_every_tuning_knob(L, knob, mode):
    """The same comparison with each tuning knob away from its default (streaming stores forced onto small arrays, no
    or rarer lockstep barriers, short and tall tiles, fewer strips per workgroup, shifted XCD bands), on shapes with
    partly filled workgroups (641 and 1409 columns: 6 and 13 strips) and with ragged last tiles."""
    key, value, default = knob
    lib = L.lib()
    L.check(lib.mg_set_tuning(key.encode(), value))
    try:
        for shape in [(65, 641), (161, 1409), (1025, 129)]:
            _fused_vs_oracle(L, shape, mode)
    finally:
        L.check(lib.mg_set_tuning(key.encode(), default))


def _fused_vs_oracle(L, shape, mode):
    import ctypes as C
    import torch
    name, sweeps, after, zero, prolong, resid, restrict, store_x = mode
    ny, nx = shape
    cny, cnx = (ny - 1) // 2 + 1, (nx - 1) // 2 + 1
    rng = np.random.default_rng(ny + 5 * nx + sweeps)
    h = 0.3
    b = rng.standard_normal(shape)
    x = np.zeros(shape) if zero else rng.standard_normal(shape)
    c = rng.standard_normal((cny, cnx))
    c[0] = c[-1] = 0
    c[:, 0] = c[:, -1] = 0
    # the oracle's sequence
    want, tmp = x.copy(), x.copy()
    for k in range(sweeps):
        if prolong and k == after:
            oracle.interpolate_add_to(c, want, 0, 0)
            tmp[:] = want
        oracle.smooth_sweep_jacobi(b, want, tmp, h)
    r_want, cb_want = np.zeros(shape), np.zeros((cny, cnx))
    if resid:
        oracle.residual(r_want, want, b, h)
    if restrict:
        oracle.coarse(r_want, cb_want, 0, 0)
    # the fused pass
    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dx, dc = padded(b), padded(x), padded(c)
    out, rr, cb = padded(x), padded(np.zeros(shape)), padded(np.zeros((cny, cnx)))
    ld, cld = db.stride(0), dc.stride(0)
    slots = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
    xin = None if zero else dx.data_ptr()
    xo = out.data_ptr() if store_x else None
    cargs = (dc.data_ptr() if prolong else None, ny, nx, ld, cny if prolong else 0, cnx if prolong else 0, cld if prolong else 0,
             1, ny - 1, 0, ny - 1, 0, sweeps)
    if restrict:
        assert after == 0
        L.check(lib.mg_smooth_fused_restrict(db.data_ptr(), xin, xo, rr.data_ptr(), *cargs, h, 1.0, slots.data_ptr(), 0,
                                             cb.data_ptr(), cny, cnx, cb.stride(0), 1, cny - 1, 1, 0, 0, 0, st))
    else:
        L.check(lib.mg_smooth_fused(db.data_ptr(), xin, xo, rr.data_ptr() if resid else None, *cargs, after, h, 1.0,
                                    slots.data_ptr() if resid else None, 1, 0, 0, st))
    if store_x:
        assert np.array_equal(out[:, :nx].cpu().numpy(), want)
    if resid:
        assert np.array_equal(rr[:, :nx].cpu().numpy(), r_want)
        nrm = C.c_double()
        L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(nrm), st))
        assert nrm.value == np.abs(r_want[1:-1, 1:-1]).max()
    if restrict:
        assert np.array_equal(cb[:, :cnx].cpu().numpy(), cb_want)


@pytest.mark.parametrize("shape", [(130, 1001), (300, 257), (600, 2051)], ids=str)
@pytest.mark.parametrize("mode", ["finest", "descent_no_x", "ascent", "nine_stage"])
@pytest.mark.parametrize("rows", [0, 8])
def test_fused_pass_two_row_ranges_in_one_launch(L, shape, mode, rows):
    """mg_fused_also_rows: the next fused pass produces a second row range in the same launch (the two ends of a
    slab whose interior rows were launched before the halo exchange).  Equal, bit for bit, to two launches; the
    setting is consumed by one call; a second range in front of the first is rejected."""
    import ctypes as C
    import torch
    ny, nx = shape
    rng = np.random.default_rng(ny + 5 * nx)
    h, oj = 0.3, -1
    x, b = rng.standard_normal(shape), rng.standard_normal(shape)
    cny, cnx = (ny - 2) // 2 + 2, (nx - 1) // 2 + 1
    c = rng.standard_normal((cny, cnx))
    flo, fhi = 1, ny - 1
    # the ends of a "slab": rows [2, a) and [e, ny - 1); odd cuts = fine centres of coarse rows (2T + oj, oj = -1)
    a, e = 2 + 2 * 9 + 1, ny - 1 - 2 * 10
    assert a % 2 == 1 and e % 2 == 1 and a < e

    def crows(j0, j1):  # coarse rows whose fine centre 2T + oj lies in [j0, j1)
        return -(-(j0 - oj) // 2), -(-(j1 - oj) // 2)

    lib = L.lib()
    st = torch.cuda.current_stream().cuda_stream
    db, dx, dc = padded(b), padded(x), padded(c)
    ld = db.stride(0)
    L.check(lib.mg_set_fused_tuning(2, rows))

    def run(ranges, also):
        out, rr, cb = padded(np.full(shape, 7.0)), padded(np.zeros(shape)), padded(np.full((cny, cnx), 3.0))
        slots = torch.zeros(L.NORM_SLOTS, dtype=torch.int64, device="cuda")
        for k, (j0, j1) in enumerate(ranges):
            if also is not None and k == 0:
                L.check(lib.mg_fused_also_rows(also[0], also[1], *(crows(*also) if mode in ("finest", "descent_no_x") else (0, 0)), ny, st))
            cs = (dc.data_ptr(), ny, nx, ld, cny, cnx, dc.stride(0))
            if mode == "finest":
                L.check(lib.mg_smooth_fused_restrict(db.data_ptr(), dx.data_ptr(), out.data_ptr(), rr.data_ptr(), *cs, j0, j1, flo, fhi,
                                                     oj, 6, h, 1.0, slots.data_ptr(), 0, cb.data_ptr(), cny, cnx, cb.stride(0),
                                                     *crows(j0, j1), 1, 0, 0, 0, st))
            elif mode == "descent_no_x":
                L.check(lib.mg_smooth_fused_restrict(db.data_ptr(), None, None, None, None, ny, nx, ld, 0, 0, 0, j0, j1, flo, fhi,
                                                     oj, 3, h, 1.0, slots.data_ptr(), 0, cb.data_ptr(), cny, cnx, cb.stride(0),
                                                     *crows(j0, j1), 1, 0, 0, 0, st))
            elif mode == "ascent":
                L.check(lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), out.data_ptr(), None, *cs, j0, j1, flo, fhi, oj, 6, 0, h,
                                            1.0, None, 1, 0, 0, st))
            else:
                L.check(lib.mg_smooth_fused(db.data_ptr(), None, out.data_ptr(), None, *cs, j0, j1, flo, fhi, oj, 9, 3, h, 1.0,
                                            None, 1, 0, 0, st))
        nrm = C.c_double()
        L.check(lib.mg_norm_read(slots.data_ptr(), C.byref(nrm), st))
        return out.cpu().numpy(), rr.cpu().numpy(), cb.cpu().numpy(), nrm.value

    try:
        two = run([(2, a), (e, ny - 1)], None)
        one = run([(2, a)], (e, ny - 1))
        for got, want in zip(one, two):
            assert np.array_equal(got, want)
        if mode != "descent_no_x":
            assert np.all(two[0][a:e] == 7.0) and not np.all(two[0][e : ny - 1] == 7.0)
        # consumed: the next call covers its own range only
        first_only = run([(2, a)], None)
        L.check(lib.mg_fused_also_rows(e, ny - 1, 0, 0, ny, st))
        L.check(lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), padded(x).data_ptr(), None, None, ny, nx, ld, 0, 0, 0, 2, 3, flo,
                                    fhi, oj, 1, 0, h, 1.0, None, 1, 0, 0, st))
        again = run([(2, a)], None)
        for got, want in zip(again, first_only):
            assert np.array_equal(got, want)
        # only the second range has rows
        empty_first = run([(2, 2)], (e, ny - 1))
        second_only = run([(e, ny - 1)], None)
        for got, want in zip(empty_first, second_only):
            assert np.array_equal(got, want)
        # rejected: a second range that does not follow the first; the failed call consumes the setting too
        L.check(lib.mg_fused_also_rows(2, a, 0, 0, ny, st))
        o = padded(x)
        assert lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), o.data_ptr(), None, None, ny, nx, ld, 0, 0, 0, e, ny - 1, flo, fhi,
                                   oj, 3, 0, h, 1.0, None, 1, 0, 0, st) != 0
        assert b"second row range" in lib.mg_last_error()
        L.check(lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), o.data_ptr(), None, None, ny, nx, ld, 0, 0, 0, e, ny - 1, flo, fhi,
                                    oj, 3, 0, h, 1.0, None, 1, 0, 0, st))
        assert lib.mg_fused_also_rows(5, 4, 0, 0, ny, st) != 0
        # a setting whose launch was never made (here: set for an array of another height) is refused by the next
        # fused call and dropped, instead of adding rows to that unrelated launch
        L.check(lib.mg_fused_also_rows(e, ny - 1, 0, 0, ny + 2, st))
        assert lib.mg_smooth_fused(db.data_ptr(), dx.data_ptr(), o.data_ptr(), None, None, ny, nx, ld, 0, 0, 0, 2, a, flo, fhi,
                                   oj, 3, 0, h, 1.0, None, 1, 0, 0, st) != 0
        assert b"never consumed" in lib.mg_last_error()
        stale_dropped = run([(2, a)], None)
        for got, want in zip(stale_dropped, first_only):
            assert np.array_equal(got, want)
    finally:
        L.check(lib.mg_set_fused_tuning(2, 0))
