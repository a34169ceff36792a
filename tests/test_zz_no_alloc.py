"""GPU: the allocation-free entry point dm_mse_loss_fwd_bwd_noalloc (include/dm_hip.h) against the existing
dm_mse_loss_fwd_bwd, through the C ABI (ctypes), on two calls in a row -- the second with a smaller input, so stale
partial sums of the first call in the library's static scratch array would show.  Same kernels, same summation order:
the results must be bit-identical.  Also: the size limit is enforced, and nothing is allocated by the call."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def test_mse_noalloc_matches_workspace_entry_point():
    from diffusion_model_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.dm_init_device() == 0
    assert L.dm_mse_noalloc_max_batch() == 4096
    for B, chw, seed in ((64, 3 * 96 * 96, 1), (5, 4 * 32 * 32, 2)):   # second call: smaller batch AND smaller sample
        g = torch.Generator(device="cuda").manual_seed(seed)
        out = torch.randn(B, chw, generator=g, device=dev)
        v = torch.randn(B, chw, generator=g, device=dev)
        w = torch.rand(B, generator=g, device=dev) / (B * chw)
        # existing entry point: caller workspace
        ws = torch.empty(L.dm_mse_workspace_bytes(B), dtype=torch.uint8, device=dev)
        loss_ref, dout_ref = torch.empty(1, device=dev), torch.empty_like(out)
        rc = L.dm_mse_loss_fwd_bwd(_ptr(out), _ptr(v), _ptr(w), 0.5, _ptr(dout_ref), _ptr(loss_ref), B, chw, _ptr(ws),
                                   ws.numel(), stream)
        assert rc == 0
        # allocation-free entry point: outputs preallocated, no workspace argument
        loss, dout = torch.full((1,), -7.0, device=dev), torch.full_like(out, -7.0)
        torch.cuda.synchronize()
        before = torch.cuda.memory_stats(dev)["allocation.all.allocated"]
        free_before = torch.cuda.mem_get_info(dev)[0]
        rc = L.dm_mse_loss_fwd_bwd_noalloc(_ptr(out), _ptr(v), _ptr(w), 0.5, _ptr(dout), _ptr(loss), B, chw, stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.cuda.memory_stats(dev)["allocation.all.allocated"] == before
        assert torch.cuda.mem_get_info(dev)[0] == free_before          # no hipMalloc behind torch's back either
        assert torch.equal(loss, loss_ref), (float(loss), float(loss_ref))
        assert torch.equal(dout, dout_ref)
        # and against the plain statement of the op (main_stl10.py:477-482), fp32 tolerance 1e-5
        want = (w[:, None] * (out - v) ** 2).sum()
        assert abs(float(loss) - float(want)) <= 1e-5 * abs(float(want))
        # loss only (dout = NULL)
        loss2 = torch.zeros(1, device=dev)
        assert L.dm_mse_loss_fwd_bwd_noalloc(_ptr(out), _ptr(v), _ptr(w), 0.5, None, _ptr(loss2), B, chw, stream) == 0
        assert torch.equal(loss2, loss_ref)
    # beyond the fixed maximum: refused with DM_ERR_TOO_LARGE (-3), nothing launched (outputs untouched)
    B = 4097
    out = torch.zeros(B, 4, device=dev)
    loss = torch.full((1,), 123.0, device=dev)
    rc = L.dm_mse_loss_fwd_bwd_noalloc(_ptr(out), _ptr(out), _ptr(out[:, 0].contiguous()), 1.0, None, _ptr(loss), B, 4, stream)
    torch.cuda.synchronize()
    assert rc == -3 and float(loss) == 123.0
    with pytest.raises(_lib.DmError, match="fixed maximum"):
        _lib.check(rc, "dm_mse_loss_fwd_bwd_noalloc")
