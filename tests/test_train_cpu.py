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
