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


def test_static_exchange_allocates_nothing_and_can_be_captured():
    """What the entry point is for: after the first call on a device it can be recorded into a hipGraph (an allocation
    or a synchronising call inside would fail the capture) and replayed; device memory in use does not change over the
    eager calls.  Where a second GPU is visible the same holds there (every device has its own copy of the array)."""
    import ctypes as C

    import torch

    from multigrid_amd._lib import check, lib
    L = lib()
    for dev in range(min(2, torch.cuda.device_count())):
        with torch.cuda.device(dev):
            m, ld = 130, 144
            v_ref = _blocks(torch, m, ld, 5 + dev).to(f"cuda:{dev}")
            v_new = v_ref.clone()
            warm = v_ref.clone()
            stage = torch.zeros(8 * m, dtype=torch.float64, device=f"cuda:{dev}")
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                s = st.cuda_stream
                check(L.mg_block_exchange_static(warm.data_ptr(), m, ld, m * ld, s))  # first call here: the address lookup
                check(L.mg_block_exchange(v_ref.data_ptr(), m, ld, m * ld, stage.data_ptr(), s))
                st.synchronize()
                check(L.mg_graph_begin(s))
                check(L.mg_block_exchange_static(v_new.data_ptr(), m, ld, m * ld, s))
                g = C.c_void_p()
                check(L.mg_graph_end(s, C.byref(g)))
                before = v_new.clone()
                st.synchronize()
                assert torch.equal(v_new, before)  # recorded, not run
                check(L.mg_graph_launch(g, s))
                st.synchronize()
                assert torch.equal(v_new, v_ref)
                # eager calls and a second replay against as many exchanges with the caller's staging (a repeated
                # exchange still moves the corner cells, which are one exchange old); free memory stays where it was
                st.synchronize()
                free1 = torch.cuda.mem_get_info(dev)[0]
                for _ in range(3):
                    check(L.mg_block_exchange_static(v_new.data_ptr(), m, ld, m * ld, s))
                st.synchronize()
                assert torch.cuda.mem_get_info(dev)[0] == free1  # (nothing else ran in between: no torch call, no graph work)
                check(L.mg_graph_launch(g, s))
                for _ in range(4):
                    check(L.mg_block_exchange(v_ref.data_ptr(), m, ld, m * ld, stage.data_ptr(), s))
                st.synchronize()
                assert torch.equal(v_new, v_ref)
                check(L.mg_graph_destroy(g))
