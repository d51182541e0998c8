"""Re-entrancy of the C ABI (SURVEY §8(b) "Threading / sync"): the reference's UI thread renders the packed
RGBA8 view (crates/brush-viewer/src/panels/scene.rs:113) while the training task runs forward+backward
(crates/brush-train/src/train.rs:232) on the same device.  Here: two host threads, two streams, both calling
libbrush_hip.so at the same time (ctypes drops the GIL for the duration of each call); every result must equal
the single-threaded one — bit for bit for the forward outputs, within the atomic-order tolerance for gradients.
"""
import threading

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_viewer_and_trainer_threads_share_the_device():
    import torch

    import brush_amd
    from brush_amd import render as R

    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    n, deg = 200_000, 2
    C = (deg + 1) ** 2
    cloud = H.synthetic_cloud(n, deg, seed=12, mean_mult=0.05)
    p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
    c = H.reference_test_camera(640, 480)
    cam_view = brush_amd.Camera([0.4, 0.1, -8.0], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    c2 = H.reference_test_camera(800, 600)
    cam_train = brush_amd.Camera(c2["position"], c2["rotation_xyzw"], c2["fov_x"], c2["fov_y"], c2["center_uv"])
    v_out = torch.randn((600, 800, 4), device=dev) / (600 * 800)

    def viewer_frame():
        img, aux = brush_amd.render_rgba8(cam_view, (640, 480), p["means"], p["log_scales"], p["quats"], p["sh"],
                                          p["raw_opac"], max_intersects=6_000_000)
        return img, aux.num_visible.clone(), aux.num_intersections.clone()

    def train_pass():
        out, aux, u = R._forward_impl(cam_train, (800, 600), p["means"], p["log_scales"], p["quats"], p["sh"],
                                      p["raw_opac"], False, 8_000_000)
        g, block = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)
        # the WHOLE compact_gid_from_isect array is kept: entries at positions >= num_intersections are unspecified
        # (allocator leftovers, the sort's last pass writes [0, I) only) and must be the ONLY ones that can differ
        I = aux.read_num_intersections()
        return out, aux.final_index.clone(), (aux.compact_gid_from_isect.clone(), I), aux.tile_bins.clone(), block

    # single-threaded references
    ref_img, ref_v, ref_i = viewer_frame()
    ref_out, ref_fin, ref_cg, ref_bins, ref_block = train_pass()
    torch.cuda.synchronize()
    assert int(ref_v.item()) > 1000 and int(ref_i.item()) > int(ref_v.item())

    iters = 12
    results = {"viewer": [], "trainer": []}
    errors = []
    start = threading.Barrier(2)

    def run(name, fn):
        try:
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                start.wait()
                for _ in range(iters):
                    results[name].append(fn())
                stream.synchronize()
        except Exception as e:  # surfaced in the main thread
            errors.append((name, repr(e)))

    ta = threading.Thread(target=run, args=("viewer", viewer_frame))
    tb = threading.Thread(target=run, args=("trainer", train_pass))
    ta.start(); tb.start(); ta.join(); tb.join()
    torch.cuda.synchronize()
    assert not errors, errors
    assert len(results["viewer"]) == iters and len(results["trainer"]) == iters
    for img, v, i in results["viewer"]:
        assert torch.equal(img, ref_img) and torch.equal(v, ref_v) and torch.equal(i, ref_i)
    scale = float(ref_block.abs().max())
    for out, fin, cg, bins, block in results["trainer"]:
        assert torch.equal(out, ref_out) and torch.equal(fin, ref_fin)
        (cg_all, I), (ref_all, ref_I) = cg, ref_cg
        assert I == ref_I and torch.equal(bins, ref_bins)
        differing = torch.nonzero(cg_all != ref_all).flatten()
        # (round 3 saw this comparison fail on the full array: the differing positions were never shown to be the tail)
        assert differing.numel() == 0 or int(differing.min()) >= I, (int(differing.min()), I, int(differing.numel()))
        assert torch.equal(cg_all[:I], ref_all[:I])
        # float atomics: summation order differs run to run even single-threaded
        assert float((block - ref_block).abs().max()) <= 1e-4 * scale
    # thread-local error slot: a failing call on one thread does not leak into the other's status
    from brush_amd import _lib

    l = _lib.lib()
    seen = {}

    def bad():
        seen["bad"] = (l.brush_radix_argsort_u32(None, None, None, None, None, 16, 33, None, 0, None),
                       l.brush_last_hip_error())

    def good():
        seen["good"] = l.brush_last_hip_error()

    t1, t2 = threading.Thread(target=bad), threading.Thread(target=good)
    t1.start(); t1.join(); t2.start(); t2.join()
    assert seen["bad"][0] == -1 and seen["good"] == 0
