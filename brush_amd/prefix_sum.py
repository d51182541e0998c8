"""prefix_sum — mirror of crates/brush-prefix-sum/src/lib.rs:17 over the HIP C ABI."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def prefix_sum(input: torch.Tensor) -> torch.Tensor:
    """Inclusive scan of a 1-D 32-bit integer device tensor (wrapping arithmetic)."""
    assert input.is_cuda, "device tensor required"
    assert input.dim() == 1 and input.dtype in (torch.int32, torch.uint32)
    x = input.contiguous()
    out = torch.empty_like(x)
    n = x.shape[0]
    if n == 0:
        return out
    l = _lib.lib()
    nbytes = C.c_size_t()
    _lib.check(l.brush_inclusive_scan_workspace_size(n, C.byref(nbytes)), "brush_inclusive_scan_workspace_size")
    ws = torch.empty(max(nbytes.value, 1), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(l.brush_inclusive_scan_u32(x.data_ptr(), out.data_ptr(), n, ws.data_ptr(), nbytes.value, stream),
                   "brush_inclusive_scan_u32")
    return out
