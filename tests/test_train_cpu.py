"""CPU checks of the caller-side training harness pieces that do not need the HIP op."""
import math

import torch
import torch.nn.functional as F


def test_separable_ssim_equals_reference_window():
    """ssim.rs:36-101: the 2-D window is outer(g, g); the harness applies it as two 1-D passes."""
    from tests.torch_trainer import Ssim

    torch.manual_seed(0)
    a, b = torch.rand(1, 37, 53, 3), torch.rand(1, 37, 53, 3)
    s = Ssim(11, 3, torch.device("cpu"))
    g = torch.tensor([math.exp(-((x - 5) ** 2) / (2.0 * 1.5 ** 2)) for x in range(11)])
    g = g / g.sum()
    w2 = torch.outer(g, g)[None, None].repeat(3, 1, 1, 1)

    def blur(x):
        return F.conv2d(x, w2, None, stride=1, padding=6, groups=3)

    x, y = a.permute(0, 3, 1, 2), b.permute(0, 3, 1, 2)
    mu_x, mu_y = blur(x), blur(y)
    s_xx = (blur(x * x) - mu_x * mu_x).clamp_min(0)
    s_yy = (blur(y * y) - mu_y * mu_y).clamp_min(0)
    s_xy = blur(x * y) - mu_x * mu_y
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    want = (((mu_x * mu_y * 2 + c1) * (s_xy * 2 + c2)) / ((mu_x * mu_x + mu_y * mu_y + c1) * (s_xx + s_yy + c2))).mean()
    got = s.ssim(a, b)
    assert abs(float(got) - float(want)) < 1e-5
    assert abs(float(s.ssim(a, a)) - 1.0) < 1e-4


def test_lazy_sh_table_rows_are_the_reciprocal_bias_corrections():
    """brush_lazy_sh_fill_table (host code of the library, no GPU): row i = (1 / (1 - b1^t), 1 / (1 - b2^t), lr, lerp)
    of optimizer time t = base + 1 + i, the floats an eager step of that time is given; argument checks."""
    import ctypes as C

    import numpy as np

    from brush_amd import _lib

    l = _lib.lib()
    base, rows = 7, 300
    tab = np.full((rows, 4), np.nan, np.float32)
    assert l.brush_lazy_sh_fill_table(0.9, 0.999, 0.004, 0.05, base, rows, tab.ctypes.data) == 0
    t = np.arange(base + 1, base + 1 + rows, dtype=np.float64)
    want1, want2 = 1.0 / (1.0 - np.float64(np.float32(0.9)) ** t), 1.0 / (1.0 - np.float64(np.float32(0.999)) ** t)
    # f32 powf, an f32 subtraction from 1 (cancellation for beta2 at small t: relative error up to eps / (1 - b2^t)) and an
    # f32 division
    eps = np.finfo(np.float32).eps
    assert np.all(np.abs(tab[:, 0] - want1) <= 4 * eps * want1 * (1.0 + want1))
    assert np.all(np.abs(tab[:, 1] - want2) <= 4 * eps * want2 * (1.0 + want2))
    assert np.all(tab[:, 2] == np.float32(0.004)) and np.all(tab[:, 3] == np.float32(0.05))
    assert np.all(np.diff(tab[:, 0]) <= 0) and np.all(np.diff(tab[:, 1]) <= 0) and tab[-1, 0] >= 1.0
    assert l.brush_lazy_sh_fill_table(0.9, 0.999, 0.004, 0.05, base, rows, None) != 0
    assert l.brush_lazy_sh_fill_table(0.9, 0.999, 0.004, 0.05, 0xFFFFFFF0, 32, tab.ctypes.data) != 0
    # the flush refuses a state whose pending times do not fit the table, and an SH degree whose rows are not whole chunks
    z = _lib.BrushLazySh()
    z.table, z.base, z.capacity, z.now = tab.ctypes.data, 0, 4, 9
    z.sh_time = z.sh_moment1 = z.sh_moment2 = tab.ctypes.data
    assert l.brush_lazy_sh_flush(C.byref(z), tab.ctypes.data, 16, 3, None) != 0
    z.now = 3
    assert l.brush_lazy_sh_flush(C.byref(z), tab.ctypes.data, 16, 2, None) != 0   # degree 2: 27 floats per row
    assert l.brush_lazy_sh_flush(C.byref(z), tab.ctypes.data, 0, 3, None) == 0    # nothing to do
