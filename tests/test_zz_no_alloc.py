"""mg_block_exchange_static, the entry point that allocates nothing: its staging lives in a device
array built into the library.  Called through the C ABI and compared bit for bit with
mg_block_exchange (caller-provided stage) -- twice in a row, the second time on smaller blocks, so
that stale staging contents of the first call would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _blocks(torch, m, ld, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(4, m, ld, dtype=torch.float64, generator=g).cuda()


def _exchange_both(torch, L, check, m, ld, seed):
    v_ref = _blocks(torch, m, ld, seed)
    v_new = v_ref.clone()
    before = v_ref.clone()
    stage = torch.full((8 * m,), float("nan"), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(L.mg_block_exchange(v_ref.data_ptr(), m, ld, m * ld, stage.data_ptr(), s))
    check(L.mg_block_exchange_static(v_new.data_ptr(), m, ld, m * ld, s))
    torch.cuda.synchronize()
    a, b = v_ref.cpu().numpy(), v_new.cpu().numpy()
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), f"m={m}: static staging differs from caller staging"
    assert not np.array_equal(a, before.cpu().numpy()), "the exchange moved nothing"
    # the reference's exchange (core/parallel_multigrid.py:83-97): rank 0's east halo column is rank 1's column 2
    assert np.array_equal(b[0][1:-1, m - 1], before.cpu().numpy()[1][1:-1, 2])


def test_static_exchange_equals_staged_exchange_twice():
    import torch

    from multigrid_amd._lib import check, lib
    L = lib()
    torch.cuda.set_device(0)
    _exchange_both(torch, L, check, 1026, 1040, seed=1)   # first call on the device: looks the array up
    _exchange_both(torch, L, check, 34, 48, seed=2)       # smaller: most of the array still holds call 1's lines
    _exchange_both(torch, L, check, 258, 258, seed=3)


def test_static_exchange_rejects_sizes_beyond_its_array():
    import torch

    from multigrid_amd import _lib
    L = _lib.lib()
    v = torch.zeros(64, dtype=torch.float64, device="cuda")
    m = _lib.BLOCK_EXCHANGE_STATIC_MAX_SIDE + 2
    assert L.mg_block_exchange_static(v.data_ptr(), m, m, m * m, None) != 0
    assert b"exceeds" in L.mg_last_error()
    assert L.mg_block_exchange_static(None, 10, 10, 100, None) != 0
