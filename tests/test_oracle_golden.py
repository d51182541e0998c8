"""Pins the CPU oracle against the reference's own golden vectors.

Mirrors `test_reference` / `renders_at_all` of crates/brush-render/src/render.rs:652-833 with the
same tolerances (render.rs:815-830).
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H


@pytest.mark.parametrize("case", ["tiny_case", "basic_case"])
def test_reference_golden(case):
    d = H.load_case(case)
    h, w, _ = d["out_img"].shape
    u = H.reference_test_uniforms(w, h, 3)
    out, aux = O.render_forward(u, d["means"], d["scales"], d["quats"], d["coeffs"], d["opacities"])
    V = int(aux["num_visible"][0])
    assert V == d["means"].shape[0]
    perm = aux["global_from_compact_gid"][:V]
    # depth order is ascending and matches the fixture's depths
    depths = d["depths"][perm]
    assert np.all(np.diff(depths) >= 0)

    def chk(name, a, b, rtol, atol):
        ok, err, bad = H.all_close_report(a, b, rtol, atol)
        assert ok, f"{case}:{name} max_abs_err={err} bad={bad}"

    chk("xys", aux["projected_splats"][:V, 0:2], d["xys"][perm], 1e-4, 1e-10)
    chk("conics", aux["projected_splats"][:V, 2:5], d["conics"][perm], 1e-4, 5e-7)
    chk("out_img", out[..., :3], d["out_img"], 1e-4, 1e-9)
    # the f64 evaluation of the same walk (arbiter of the GPU pixel tolerance) agrees with the pinned f32 result
    exact, cond = O.rasterize_forward_f64(u, aux)
    assert np.isfinite(cond).all() and cond.max() < 1e4  # well-conditioned fixture
    assert np.abs(exact - out.astype(np.float64)).max() < 2e-6
    chk("out_img_f64", exact[..., :3].astype(np.float32), d["out_img"], 1e-4, 1e-9)

    # loss = mean((rgb - crab)^2)  (render.rs:786-789)
    v_out = np.zeros((h, w, 4), np.float32)
    v_out[..., :3] = 2.0 * (out[..., :3] - H.crab_rgb()) / (h * w * 3)
    assert np.abs(v_out[..., :3] - d["v_out_img"]).max() < 1e-9
    g = O.render_backward(u, aux, d["means"], d["scales"], d["quats"], d["opacities"], out, v_out)
    chk("v_xy", g["v_xy"], d["v_xy"], 1e-4, 1e-9)
    chk("v_opacities", g["v_opac"], d["v_opacities"], 1e-4, 1e-10)
    chk("v_coeffs", g["v_sh"], d["v_coeffs"], 1e-4, 1e-9)
    chk("v_scales", g["v_scales"], d["v_scales"], 1e-4, 1e-9)
    chk("v_means", g["v_means"], d["v_means"], 1e-4, 1e-9)
    chk("v_quats", H.unnormalised_quat_grad(d["quats"], g["v_quats"]), d["v_quats"], 1e-1, 1e-1)
    # not asserted by the reference, but the fixture holds it:
    chk("v_conics", g["v_conics"], d["v_conics"][perm], 1e-4, 1e-8)


def test_renders_at_all():
    """render.rs:652-693: 8 degenerate splats at the camera origin are all culled."""
    n = 8
    u = O.make_uniforms([0, 0, 0], [0, 0, 0, 1], 0.5, 0.5, [0.5, 0.5], [32, 32], 0)
    means = np.zeros((n, 3), np.float32)
    log_scales = np.full((n, 3), 2.0, np.float32)
    quats = np.tile(np.array([[0, 0, 0, 1]], np.float32), (n, 1))  # glam IDENTITY.to_array()
    sh = np.ones((n, 1, 3), np.float32)
    raw = np.zeros(n, np.float32)
    out, aux = O.render_forward(u, means, log_scales, quats, sh, raw)
    assert int(aux["num_visible"][0]) == 0
    assert abs(out[..., :3].mean()) < 1e-5
    assert out[..., 3].mean() == 0.0
    g = O.render_backward(u, aux, means, log_scales, quats, raw, out, np.ones_like(out))
    for k in ("v_means", "v_xy", "v_scales", "v_quats", "v_sh", "v_opac"):
        assert not g[k].any()


def test_raster_u32_matches_float():
    """rasterize.wgsl:106-109: packed RGBA8 = trunc(clamp(x*255, 0, 255)), little-endian."""
    d = H.load_case("basic_case")
    h, w, _ = d["out_img"].shape
    u = H.reference_test_uniforms(w, h, 3)
    out, _ = O.render_forward(u, d["means"], d["scales"], d["quats"], d["coeffs"], d["opacities"])
    packed, _ = O.render_forward(u, d["means"], d["scales"], d["quats"], d["coeffs"], d["opacities"],
                                 raster_u32=True)
    want = np.clip(out * 255.0, 0, 255).astype(np.uint32)
    want = want[..., 0] | (want[..., 1] << 8) | (want[..., 2] << 16) | (want[..., 3] << 24)
    assert np.array_equal(packed, want)
