"""radix_argsort — mirror of crates/brush-sort/src/lib.rs:32-37 over the HIP C ABI."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def radix_argsort(input_keys: torch.Tensor, input_values: torch.Tensor, n_sort: torch.Tensor,
                  sorting_bits: int):
    """Stable argsort of the first `n_sort` (device scalar) pairs on the low bits of the keys.

    Same contract as the reference: keys and values are 1-D 32-bit integer tensors of equal
    length (assert, lib.rs:38), `sorting_bits <= 32` (assert, lib.rs:39), `n_sort` lives on the
    device and is never read back.  Returns (keys, values) as new tensors.
    """
    assert input_keys.shape[0] == input_values.shape[0]
    assert sorting_bits <= 32
    assert input_keys.is_cuda and input_values.is_cuda and n_sort.is_cuda, "device tensors required"
    assert input_keys.dtype in (torch.int32, torch.uint32) and input_values.dtype in (torch.int32, torch.uint32)
    keys = input_keys.contiguous()
    vals = input_values.contiguous()
    n_sort = n_sort.reshape(-1)[:1].contiguous()
    max_n = keys.shape[0]
    out_k = torch.empty_like(keys)
    out_v = torch.empty_like(vals)
    l = _lib.lib()
    nbytes = C.c_size_t()
    _lib.check(l.brush_radix_argsort_workspace_size(max_n, C.byref(nbytes)), "brush_radix_argsort_workspace_size")
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=keys.device)
    with torch.cuda.device(keys.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(l.brush_radix_argsort_u32(keys.data_ptr(), vals.data_ptr(), out_k.data_ptr(), out_v.data_ptr(),
                                             n_sort.data_ptr(), max_n, int(sorting_bits), ws.data_ptr(),
                                             nbytes.value, stream), "brush_radix_argsort_u32")
    return out_k, out_v
