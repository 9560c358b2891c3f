"""The allocation-free masked-loss entry point (mmae_masked_loss_static: partial sums in the library's own fixed device
array) against mmae_masked_loss with a caller workspace, through the C ABI: same kernels, so results are bitwise equal.
Two calls in a row, the second with a smaller input; sizes beyond the fixed array are refused with a message."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bind():
    from multimae_amd import _lib
    lib = _lib.load()
    return lib, lib.mmae_masked_loss_static


def _case(lib, fn, B, Cc, S, patch, kind, norm_pix, bf16, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    pred = torch.randn(B, Cc, S, S, device="cuda", generator=g)
    if bf16:
        pred = pred.bfloat16()
    target = torch.randn(B, Cc, S, S, device="cuda", generator=g)
    mask = (torch.rand(B, (S // patch) ** 2, device="cuda", generator=g) < 0.7).long()
    mask[0] = 0  # a sample without masked tokens: dropped by the nanmean, NaN gradient rows as in the reference
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    # reference: the workspace entry point
    loss_a = torch.empty((), device="cuda")
    d_a = torch.empty_like(pred)
    nbytes = lib.mmae_masked_loss_workspace_bytes(B, Cc, S, S, patch)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    assert lib.mmae_masked_loss(kind, int(norm_pix), ptr(pred), int(bf16), ptr(target), ptr(mask), B, Cc, S, S, patch, ptr(loss_a),
                                ptr(d_a), ptr(ws), nbytes, stream) == 0, lib.mmae_last_error()
    loss_b = torch.full((), -1.0, device="cuda")
    d_b = torch.full_like(pred, 7.0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rc = fn(kind, int(norm_pix), ptr(pred), int(bf16), ptr(target), ptr(mask), B, Cc, S, S, patch, ptr(loss_b), ptr(d_b), stream)
    assert rc == 0, lib.mmae_last_error()
    torch.cuda.synchronize()
    return loss_a, d_a, loss_b, d_b, free0 - torch.cuda.mem_get_info()[0]


def test_static_scratch_masked_loss_equals_the_workspace_entry_point():
    lib, fn = _bind()
    # first call (resolves the array's address on this device), then a smaller input that reuses the array's front
    first = _case(lib, fn, B=16, Cc=3, S=224, patch=16, kind=0, norm_pix=True, bf16=True, seed=1)
    second = _case(lib, fn, B=5, Cc=1, S=64, patch=16, kind=1, norm_pix=False, bf16=False, seed=2)
    for n, (loss_a, d_a, loss_b, d_b, grew) in enumerate((first, second)):
        assert torch.isfinite(loss_a).item() and torch.equal(loss_a, loss_b), (n, loss_a, loss_b)
        assert torch.equal(d_a.view(torch.int16 if d_a.dtype == torch.bfloat16 else torch.int32),
                           d_b.view(torch.int16 if d_b.dtype == torch.bfloat16 else torch.int32)), n  # bitwise, NaN rows included
        if n == 1:
            assert grew == 0, f"device memory in use grew by {grew} bytes across the call"
    # beyond the fixed array: refused before any launch
    ptr = C.c_void_p(first[1].data_ptr())
    rc = fn(0, 0, ptr, 1, ptr, None, 1 << 20, 3, 224, 224, 16, ptr, None, None)
    assert rc == -3 and b"fixed array" in lib.mmae_last_error()
    # tensors that are not on the current device (here: a host buffer for the loss) are refused before any launch
    host_loss = (C.c_float * 1)()
    rc = fn(0, 0, ptr, 1, ptr, None, 4, 3, 224, 224, 16, C.cast(host_loss, C.c_void_p), None, None)
    assert rc == -1 and b"masked_loss_static" in lib.mmae_last_error()
