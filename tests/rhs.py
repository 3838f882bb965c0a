"""Right-hand sides used by the reference's drivers, rebuilt with the same numpy expressions
so the arrays are bit-identical to the ones the golden vectors were computed from."""
import numpy as np


def _grid(n):
    nx0 = 2 ** (n + 1) + 1
    x = np.linspace(-10, 10, nx0)
    y = np.linspace(-10, 10, nx0)
    h = y[1] - y[0]
    X, Y = np.meshgrid(x, y)
    return X, Y, h


def test_rhs(n):
    """Eight alternating vortices, test/test.py:40-58."""
    X, Y, h = _grid(n)
    b0 = np.zeros_like(X)
    sign = 1
    for xv, yv in zip([4, -2, 6, 3, 3, 5, -5, -7], [-2, 6, 6, -5, 5, -5, -2, -2]):
        r = (X - xv) ** 2 + (Y - yv) ** 2
        b0 += sign * 50 * np.exp(-r * 7)
        sign *= -1
    return b0, h


test_rhs.__test__ = False  # not a pytest test


def main_rhs(n):
    """Single Gaussian, core/parallel_multigrid.py:359-372."""
    X, Y, h = _grid(n)
    r = X ** 2 + Y ** 2
    return 50 * np.exp(-r * 4), h
