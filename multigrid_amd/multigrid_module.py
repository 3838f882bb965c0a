"""GPU drop-in for the reference's compiled kernel module ``multigrid_module``.

Same eight functions, same positional arguments, same in-place semantics and ``None`` return as
the exports of core/numba_modules.py:10-180 (reference checkout), each executed by a
hand-written HIP kernel of libmg_hip.so through the C ABI of include/mg_hip.h:

    split, laplacian, residual, smooth_sweep_jacobi, smooth, smooth_altern, coarse,
    interpolate_add_to

Arguments may be

* 2-D float64 ``torch`` tensors on an MI355X with unit column stride (any row stride): the kernel
  works on them in place on the current torch stream, nothing is copied;
* 2-D float64 ``numpy`` arrays: copied to the GPU, processed there, and the arrays the reference
  writes are copied back, so host code written against the reference keeps working unchanged.

As in the numba build a wrong dtype or rank raises ``TypeError``.  Unlike it, shapes are checked
(a reference call that would index out of bounds raises ``MgError`` here).
Results are bit-identical to the reference's for the same inputs.
"""
import numpy as np
import torch

from ._lib import (SOR_LEXICOGRAPHIC, SOR_REDBLACK, MgError, check, lib)

__all__ = ["split", "laplacian", "residual", "smooth_sweep_jacobi", "smooth", "smooth_altern", "coarse",
           "interpolate_add_to"]

_ORDERINGS = {"lexicographic": SOR_LEXICOGRAPHIC, "redblack": SOR_REDBLACK}


def _device():
    if not torch.cuda.is_available():
        raise MgError("no GPU visible: multigrid_amd runs on MI355X only and has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Args:
    """Maps the arguments of one call to device tensors and writes results back to numpy arrays."""

    def __init__(self):
        self._seen = {}
        self._writeback = []
        self.device = None

    def __call__(self, a, name, out=False):
        if isinstance(a, torch.Tensor):
            if a.dtype != torch.float64 or a.dim() != 2:
                raise TypeError(f"{name}: expected a 2-D float64 array, got {a.dtype} with {a.dim()} dims")
            if not a.is_cuda:
                raise TypeError(f"{name}: torch tensors must live on the GPU (pass numpy arrays for host data)")
            if a.shape[1] > 1 and a.stride(1) != 1:
                raise TypeError(f"{name}: unit column stride required")
            t = a
        elif isinstance(a, np.ndarray):
            if a.dtype != np.float64 or a.ndim != 2:
                raise TypeError(f"{name}: expected a 2-D float64 array, got {a.dtype} with {a.ndim} dims")
            key = id(a)
            if key in self._seen:
                t = self._seen[key]
            else:
                t = torch.from_numpy(np.ascontiguousarray(a)).to(_device())
                self._seen[key] = t
            if out and all(x is not a for x, _ in self._writeback):
                self._writeback.append((a, t))
        else:
            raise TypeError(f"{name}: expected numpy.ndarray or torch.Tensor, got {type(a).__name__}")
        if self.device is None:
            self.device = t.device
        elif t.device != self.device:
            raise MgError(f"{name}: arrays live on different devices")
        return t

    def finish(self):
        for a, t in self._writeback:
            a[...] = t.cpu().numpy()


def _ld(t):
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _same_shape(name, *ts):
    s = ts[0].shape
    for t in ts[1:]:
        if t.shape != s:
            raise MgError(f"{name}: arrays of different shapes {tuple(s)} vs {tuple(t.shape)}")
    ld = _ld(ts[0])
    for t in ts[1:]:
        if _ld(t) != ld:
            raise MgError(f"{name}: arrays must share one row stride")
    return s[0], s[1], ld


def split(A_in, A_out, rank):
    """core/numba_modules.py:10-20 -- copy the corner block of ``A_in`` that belongs to ``rank``."""
    g = _Args()
    a, o = g(A_in, "A_in"), g(A_out, "A_out", out=True)
    if o.shape[0] != o.shape[1]:
        raise MgError("split: A_out must be square")
    with torch.cuda.device(g.device):
        check(lib().mg_split(a.data_ptr(), a.shape[0], a.shape[1], _ld(a), o.data_ptr(), o.shape[0], _ld(o),
                             int(rank), _stream()))
    g.finish()


def laplacian(a, lplc, h=1):
    """core/numba_modules.py:23-32 -- 5-point laplacian on the interior."""
    g = _Args()
    ta, tl = g(a, "a"), g(lplc, "lplc", out=True)
    ny, nx, ld = _same_shape("laplacian", ta, tl)
    with torch.cuda.device(g.device):
        check(lib().mg_laplacian(ta.data_ptr(), tl.data_ptr(), ny, nx, ld, float(h), _stream()))
    g.finish()


def residual(r, a, b, h):
    """core/numba_modules.py:35-50 -- r = b - laplacian(a) on the interior."""
    g = _Args()
    tr, ta, tb = g(r, "r", out=True), g(a, "a"), g(b, "b")
    ny, nx, ld = _same_shape("residual", tr, ta, tb)
    with torch.cuda.device(g.device):
        check(lib().mg_residual(tr.data_ptr(), ta.data_ptr(), tb.data_ptr(), ny, nx, ld, 1, ny - 1, float(h), None,
                                1, 0, 0, _stream()))
    g.finish()


def smooth_sweep_jacobi(b, a, a_temp, h):
    """core/numba_modules.py:53-66 -- Jacobi sweep into ``a_temp`` (interior), then ALL of
    ``a_temp`` is copied to ``a`` as the reference's second loop does."""
    g = _Args()
    tb, ta, tt = g(b, "b"), g(a, "a", out=True), g(a_temp, "a_temp", out=True)
    ny, nx, ld = _same_shape("smooth_sweep_jacobi", tb, ta, tt)
    with torch.cuda.device(g.device):
        L = lib()
        check(L.mg_jacobi(tb.data_ptr(), ta.data_ptr(), tt.data_ptr(), ny, nx, ld, 1, ny - 1, float(h), 1.0, 1, 0, 0,
                          _stream()))
        check(L.mg_copy2d(ta.data_ptr(), ld, tt.data_ptr(), ld, ny, nx, _stream()))
    g.finish()


def _sor(name, b, x, h, r, iterations, backward, ordering):
    g = _Args()
    tb, tx, tr = g(b, "b"), g(x, "x", out=True), g(r, "r", out=True)
    ny, nx, ld = _same_shape(name, tb, tx, tr)
    with torch.cuda.device(g.device):
        check(lib().mg_sor(tb.data_ptr(), tx.data_ptr(), tr.data_ptr(), ny, nx, ld, float(h), int(iterations),
                           backward, _ORDERINGS[ordering], 1, 0, _stream()))
    g.finish()


def smooth(b, x, h, r, iterations, ordering="lexicographic"):
    """core/numba_modules.py:69-95 -- ``iterations`` forward SOR sweeps (omega 1.5), then the residual."""
    _sor("smooth", b, x, h, r, iterations, 0, ordering)


def smooth_altern(b, x, h, r, iterations, ordering="lexicographic"):
    """core/numba_modules.py:98-129 -- ``iterations`` x (forward + backward SOR sweep), then the
    residual.  ``ordering="redblack"`` selects the two-colour ordering (not a reference ordering)."""
    _sor("smooth_altern", b, x, h, r, iterations, 1, ordering)


def coarse(a, a_crs, ofst_i=0, ofst_j=0):
    """core/numba_modules.py:132-160 -- full-weighting restriction of ``a`` into the interior of ``a_crs``."""
    g = _Args()
    tf, tc = g(a, "a"), g(a_crs, "a_crs", out=True)
    with torch.cuda.device(g.device):
        check(lib().mg_restrict(tf.data_ptr(), tf.shape[0], tf.shape[1], _ld(tf), tc.data_ptr(), tc.shape[0],
                                tc.shape[1], _ld(tc), 1, tc.shape[0] - 1, int(ofst_i), int(ofst_j), 1, 0, 0,
                                _stream()))
    g.finish()


def interpolate_add_to(a, a_new, ofst_i=0, ofst_j=0):
    """core/numba_modules.py:163-180 -- bilinear interpolation of ``a`` added into ``a_new``."""
    g = _Args()
    tc, tf = g(a, "a"), g(a_new, "a_new", out=True)
    with torch.cuda.device(g.device):
        check(lib().mg_prolong_add(tc.data_ptr(), tc.shape[0], tc.shape[1], _ld(tc), tf.data_ptr(), tf.shape[0],
                                   tf.shape[1], _ld(tf), 0, tf.shape[0], 0, tf.shape[1] - 1, int(ofst_i), int(ofst_j),
                                   1, 0, 0, _stream()))
    g.finish()
