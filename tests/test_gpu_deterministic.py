"""Deterministic mode (BrushAux::flags & BRUSH_AUX_DETERMINISTIC; SURVEY §7 "offer a deterministic (sorted
segmented-reduce) mode"): the compositing backward stores one gradient row per intersection and the rows are summed per
splat in a fixed order, so gradients are bitwise reproducible run to run (the default uses hardware float atomics, the
reference a CAS queue, rasterize_backwards.wgsl:276-301: both depend on arrival order).

The mode is an ARGUMENT of every call (no process-wide switch), so ONE interpreter runs both modes here: the
oracle-parity tests of test_gpu_render.py / test_gpu_train.py are called again with the flag set, next to calls of the
default mode on the same device, plus the reproducibility checks below."""
import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import brush_amd.render as R

    R.DEBUG_POISON = True
    return torch.device("cuda:0")


@pytest.fixture
def det_default():
    """Deterministic gradients as the default of brush_amd.render for the duration of one test."""
    import brush_amd.render as R

    old = R.DETERMINISTIC
    R.DETERMINISTIC = True
    yield
    R.DETERMINISTIC = old


def _cloud_and_camera(dev, n=150000, w=640, h=480, deg=3):
    import torch

    import brush_amd

    cloud = H.synthetic_cloud(n, deg, seed=4, mean_mult=0.05)  # long tile lists, splats spanning many tiles and chunks
    p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    return p, cam


@pytest.mark.timeout(600)
def test_gradients_are_bitwise_reproducible_and_modes_coexist(dev):
    """Deterministic and default calls interleaved in one process on one device: the deterministic ones are bitwise
    reproducible, the default ones agree with them to float-atomic rounding, the forward state is identical."""
    import torch

    from brush_amd import _lib, dist as BD, render as R

    n, w, h, deg = 150000, 640, 480, 3
    C = (deg + 1) ** 2
    p, cam = _cloud_and_camera(dev, n, w, h, deg)
    v_out = torch.randn((h, w, 4), device=dev) / (h * w)
    blocks, atomics = [], []
    for rep in range(3):
        out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False,
                                      4_000_000, deterministic=True)
        assert aux.deterministic and aux.isect_unsorted_pos is not None and aux.flags == _lib.AUX_DETERMINISTIC
        g, block = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)
        # the default mode right beside it (its own aux, no isect_unsorted_pos, smaller workspace)
        out2, aux2, u2 = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"],
                                         False, 4_000_000, deterministic=False)
        assert not aux2.deterministic and aux2.isect_unsorted_pos is None and aux2.flags == 0
        g2, block2 = R._backward_impl(u2, aux2, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out2, v_out)
        torch.cuda.synchronize()
        assert torch.equal(out, out2) and torch.equal(aux.final_index, aux2.final_index)
        assert torch.equal(aux.compact_gid_from_isect[:aux.read_num_intersections()],
                           aux2.compact_gid_from_isect[:aux2.read_num_intersections()])
        blocks.append(block.clone())
        atomics.append(block2.clone())
    I = aux.read_num_intersections()
    pos = aux.isect_unsorted_pos[:I].long()
    assert int(torch.bincount(pos, minlength=I).max()) == 1 and int(pos.max()) == I - 1  # a permutation of 0..I-1
    assert torch.equal(blocks[0], blocks[1]) and torch.equal(blocks[0], blocks[2]), "gradients must be bitwise reproducible"
    assert bool(blocks[0].abs().sum() > 0)
    scale = float(blocks[0].abs().max())
    assert float((blocks[0] - atomics[0]).abs().max()) <= 1e-4 * scale
    # a flag the library does not know is refused, as is the deterministic flag without its buffer
    from brush_amd import _lib as L
    import ctypes as Cc

    s = aux2._as_struct()
    s.flags = 4
    nb = Cc.c_size_t()
    assert L.lib().brush_bwd_workspace_size_flags(n, w, h, deg, aux2.max_intersects, 4, Cc.byref(nb)) != 0
    s.flags = L.AUX_DETERMINISTIC  # isect_unsorted_pos is NULL in aux2
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    rc = L.lib().brush_render_backward(Cc.byref(u2), Cc.byref(s), p["means"].data_ptr(), p["log_scales"].data_ptr(),
                                       p["quats"].data_ptr(), p["raw_opac"].data_ptr(), n, out2.data_ptr(),
                                       v_out.data_ptr(), g2["v_means"].data_ptr(), g2["v_xy"].data_ptr(),
                                       g2["v_scales"].data_ptr(), g2["v_quats"].data_ptr(), g2["v_sh"].data_ptr(),
                                       g2["v_opac"].data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -1  # BRUSH_ERR_INVALID_ARG

    # the record form (multi-GPU path) consumes the same sums
    x = BD.ViewExchange(n, C, dev, packed=True)
    x.begin(aux)
    x.backward_records(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], out, v_out)
    r1 = x.gather()[0].clone()
    x.backward_records(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], out, v_out)
    r2 = x.gather()[0].clone()
    V = aux.read_num_visible()
    assert r1.shape[0] == V and torch.equal(r1, r2)
    grads, red = x.reduce_dense(p["means"])
    layout, _ = R.grad_block_layout(n, C)
    for name in ("v_means", "v_scales", "v_quats", "v_opac"):
        off, sz = layout[name]
        assert torch.equal(red[off:off + sz], blocks[0][off:off + sz]), name  # same sums, same VJP, 0 + x = x
    # intersection capacity overflow: truncated lists, still reproducible and finite
    out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False,
                                  100_000, deterministic=True)
    assert int(aux.overflow.item()) == 1
    a = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)[1].clone()
    b = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)[1].clone()
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())


@pytest.mark.timeout(600)
def test_training_trajectory_is_bitwise_reproducible(dev, det_default):
    import torch

    import brush_amd

    n, w, h, deg = 150000, 640, 480, 3
    p, cam = _cloud_and_camera(dev, n, w, h, deg)

    def train_once():
        torch.manual_seed(0)
        splats = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
        tr = brush_amd.SplatTrainer(splats, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0))
        gt = torch.rand((h, w, 3), device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        for _ in range(3):
            tr.step(splats, cam, gt, 1.0)
        torch.cuda.synchronize()
        return [q.detach().clone() for q in (splats.means, splats.sh_coeffs, splats.rotation, splats.raw_opacity,
                                             splats.log_scales)]

    t1, t2 = train_once(), train_once()
    assert all(torch.equal(a, b) for a, b in zip(t1, t2)), "training trajectory must be bitwise reproducible"


def _parity_cases():
    """(test function of the default-mode suites, kwargs): the oracle-parity tests re-run with the flag set."""
    from tests import test_gpu_render as TR
    from tests import test_gpu_train as TT

    cases = []
    for args in [(2000, 123, 82, 3, 0.002), (20000, 256, 192, 0, 0.01), (20000, 200, 120, 1, 0.01),
                 (20000, 200, 120, 2, 0.01), (20000, 200, 120, 4, 0.01), (150000, 640, 480, 3, 0.05),
                 (300000, 800, 800, 0, 1.0)]:
        cases.append((TR.test_matches_oracle, dict(zip(("n", "w", "h", "deg", "mult"), args))))
    for case in ("tiny_case", "basic_case"):
        cases.append((TR.test_reference_golden, dict(case=case)))
    for n in (0, 1, 2, 65):
        cases.append((TR.test_tiny_and_empty_inputs, dict(n=n)))
    for f in (TR.test_intersection_overflow_is_flagged, TR.test_walk_queue_overflow_falls_back_inline,
              TR.test_depth_ties_are_deterministic, TR.test_headline_size_matches_oracle, TR.test_alpha_clamp_quirk,
              TR.test_training_steps_reduce_loss, TT.test_hip_trainer_tracks_torch_trainer,
              TT.test_training_with_refinement_runs):
        cases.append((f, {}))
    for n, deg in ((3000, 2), (3001, 3), (1026, 0)):
        cases.append((TT.test_fused_backward_adam_equals_separate_calls, dict(n=n, deg=deg)))
    return cases


def _case_id(c):
    return c[0].__name__ + ("[" + "-".join(str(v) for v in c[1].values()) + "]" if c[1] else "")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", _parity_cases(), ids=_case_id)
def test_oracle_parity_holds_in_deterministic_mode(dev, det_default, case):
    """The same oracle-parity tests as the default mode (forward state bit-exact, gradients against the f64 arbiter,
    trainer against its torch restatement), in this process, with deterministic gradients selected per call."""
    fn, kwargs = case
    fn(dev, **kwargs)


def test_c3_reference_cap_in_deterministic_mode(dev, det_default):
    from tests import test_gpu_render as TR

    cloud = H.synthetic_cloud(3_000_000, 3, seed=4, mean_mult=0.12)
    TR.test_c3_scale_reference_cap_overflows(dev, cloud)
