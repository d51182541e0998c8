"""GPU check of the compact gradient exchange end to end: two ranks share the one GPU of the test box
and talk over gloo (RCCL refuses two ranks on one device); every rank runs the HIP op on its own view,
packs, all-gathers and expands with the HIP kernels, and the result is compared with the dense sum of
the two views' gradient blocks.  The second and third exchange go through the optimistic sizing path
(previous size + 12.5 %): once with a view that outgrew the hint (redo), once within it."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, chunks):
    os.environ["BRUSH_EXCHANGE_CHUNKS"] = str(chunks)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import brush_amd
        from brush_amd import dist as BD
        from brush_amd import render as R

        dev = torch.device("cuda:0")
        n, w, h, deg = 40000, 320, 200, 2
        C = (deg + 1) ** 2
        cloud = H.synthetic_cloud(n, deg, seed=17, mean_mult=0.003)
        p = {k: torch.from_numpy(v).to(dev) for k, v in cloud.items()}
        c = H.reference_test_camera(w, h)
        results = []
        # step 0: narrow views; step 1: the camera backs off (more splats visible than hint allows); step 2: same
        for step, z in enumerate((-6.0, -30.0, -30.0)):
            cam = brush_amd.Camera([0.5 * rank, -0.3 * rank, z], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
            out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"],
                                          False, 4_000_000)
            torch.manual_seed(100 + step)
            v_out = torch.randn((h, w, 4), device=dev) / (h * w)
            g, block = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)
            pf = BD.param_grad_floats(n, C)
            dense = block[:pf].detach().cpu().clone()
            dist.all_reduce(dense)                         # reference: dense sum over the two views (CPU gloo)
            hint_before = dict(BD._ROWS_HINT)
            BD.allreduce_param_grads_compact(block, aux, p["means"], n, C)
            got = block[:pf].detach().cpu()
            err = float((got.double() - dense.double()).abs().max())
            results.append((step, aux.read_num_visible(), err, float(dense.abs().max()), bool(hint_before)))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chunks", [1, 2])  # 2 = the overlapped two-half exchange used from 4 views on
def test_compact_exchange_two_ranks_one_gpu(chunks):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, chunks)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, res = q.get(timeout=500)
        got[rank] = res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        steps = got[rank]
        assert [s[0] for s in steps] == [0, 1, 2]
        assert not steps[0][4] and steps[1][4] and steps[2][4]      # hint used from the second exchange on
        assert steps[1][1] > 1.3 * steps[0][1]                      # step 1 really outgrew the hint
        for step, V, err, scale, _ in steps:
            assert V > 1000
            assert err <= 2e-6 * scale, (rank, step, err, scale)
