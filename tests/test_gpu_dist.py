"""GPU check of the view-sharded gradient exchange end to end: two ranks share the one GPU of the test box and talk
over gloo (RCCL refuses two ranks on one device).  Every rank runs the HIP op on its own view, writes its 64-byte
records (brush_render_backward_records), all-gathers them and reduces them with the HIP kernel; the result is compared
with the dense sum of the two views' gradient blocks, must be BIT-IDENTICAL on the two ranks, and the fused
sum-into-Adam form must leave the same parameters on both ranks.  The camera backs off between steps so that a view
outgrows the records buffer after its backward was enqueued (the re-run path)."""
import math
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ctypes as C

        import brush_amd
        from brush_amd import _lib
        from brush_amd import dist as BD
        from brush_amd import render as R

        dev = torch.device("cuda:0")
        n, w, h, deg = 40000, 320, 200, 2
        ncoef = (deg + 1) ** 2
        cloud = H.synthetic_cloud(n, deg, seed=17, mean_mult=0.003)
        p = {k: torch.from_numpy(v).to(dev) for k, v in cloud.items()}
        c = H.reference_test_camera(w, h)
        xchg = BD.ViewExchange(n, ncoef, dev)
        results = []
        # step 0: narrow views; step 1: the camera backs off (far more splats visible than the buffer holds); step 2: same
        for step, z in enumerate((-6.0, -30.0, -30.0)):
            cam = brush_amd.Camera([0.5 * rank, -0.3 * rank, z], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
            out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"],
                                          False, 4_000_000)
            xchg.begin(aux)
            torch.manual_seed(100 + step)
            v_out = torch.randn((h, w, 4), device=dev) / (h * w)
            # reference: dense gradients of this view, summed over the two views on the CPU
            g, block = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], ncoef, out, v_out)
            pf = BD.param_grad_floats(n, ncoef)
            dense = block[:pf].detach().cpu().clone()
            dist.all_reduce(dense)
            xchg.backward_records(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], out, v_out)
            recs = xchg.gather()
            grads, red = xchg.reduce_dense(p["means"])
            torch.cuda.synchronize()
            got = red[:pf].detach().cpu()
            # the HIP record kernel against the torch restatement fed with the HIP dense gradients
            V = aux.read_num_visible()
            mine = recs[rank]  # default (padded) form: [W, rows, 16], the first V rows of a view are its records
            assert mine.shape[0] >= V and recs.shape[0] == world and not xchg.packed
            want_rec = BD.records_from_dense_torch(g, aux, n, (w, h), V)
            rec_err = float((mine[:V, 1:].double() - want_rec[:V, 1:].double()).abs().max()
                            / (want_rec[:V, 1:].abs().max() + 1e-30))
            gid_ok = bool(torch.equal(mine[:V, 0].contiguous().view(torch.int32),
                                      want_rec[:V, 0].contiguous().view(torch.int32)))
            results.append(dict(step=step, V=V, err=float((got.double() - dense.double()).abs().max()),
                                scale=float(dense.abs().max()), rec_err=rec_err, gid_ok=gid_ok,
                                regrown=xchg.regrown, rows=int(xchg._rows), red=got.numpy()))
        # fused form: the same records straight into Adam == brush_adam_step on the reduced dense gradients
        params = {k: p[k].clone() for k in ("means", "log_scales", "quats", "raw_opac", "sh")}
        m1 = torch.zeros(n * (11 + 3 * ncoef), device=dev)
        m2 = torch.zeros_like(m1)
        cfg = _lib.BrushAdamConfig(1.6e-4, 0.01, 0.002, 0.05, 0.004, 1.0 / 20.0, 0.9, 0.999, 1e-15, 1, 1)
        ref = {k: v.clone() for k, v in params.items()}
        rm1, rm2 = torch.zeros_like(m1), torch.zeros_like(m2)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().brush_adam_step(C.byref(cfg), n, deg, ref["means"].data_ptr(), ref["log_scales"].data_ptr(),
                                                  ref["quats"].data_ptr(), ref["raw_opac"].data_ptr(), ref["sh"].data_ptr(),
                                                  grads["v_means"].data_ptr(), grads["v_scales"].data_ptr(),
                                                  grads["v_quats"].data_ptr(), grads["v_opac"].data_ptr(),
                                                  grads["v_sh"].data_ptr(), rm1.data_ptr(), rm2.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "brush_adam_step")
        acc, cnt = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        xchg.reduce_adam(cfg, (w, h), params["means"], params["log_scales"], params["quats"], params["raw_opac"],
                         params["sh"], m1, m2, None, acc, cnt)
        torch.cuda.synchronize()
        adam = {k: float((params[k].double() - ref[k].double()).abs().max()) for k in params}
        moved = {k: float((params[k].double() - p[k].double()).abs().max()) for k in params}
        q.put((rank, results, adam, moved, {k: v.cpu().numpy() for k, v in params.items()},
               acc.cpu().numpy(), cnt.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_record_exchange_two_ranks_one_gpu():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=500)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        steps = got[rank][0]
        assert [s["step"] for s in steps] == [0, 1, 2]
        assert steps[1]["V"] > 1.3 * steps[0]["V"]                 # step 1 really outgrew the buffer ...
        assert steps[1]["regrown"] >= 1 and steps[2]["regrown"] == steps[1]["regrown"]  # ... once
        for s in steps:
            assert s["V"] > 1000 and s["gid_ok"]
            # the records and the dense gradients come from two backward calls whose float atomics land in different
            # orders (+ one extra rounding in v_rgb = v_sh0 / Y0): a few ulp of the tensor's scale
            assert s["rec_err"] <= 5e-6
            assert s["err"] <= 5e-6 * s["scale"], (rank, s["step"], s["err"], s["scale"])
    # the reduced gradients are the same bits on both ranks, every step
    for a, b in zip(got[0][0], got[1][0]):
        assert np.array_equal(a["red"].view(np.uint32), b["red"].view(np.uint32)), a["step"]
    # fused Adam: equal to brush_adam_step on the reduced gradients (same formulas, other kernel), really moved the
    # parameters, and left the same bits on both ranks; statistics = two views' worth
    for rank in range(world):
        adam, moved = got[rank][1], got[rank][2]
        for k in adam:
            # (an ulp of the parameter itself: the two kernels round the quaternion chain rule differently)
            assert moved[k] > 0 and adam[k] <= 1e-5 * moved[k] + 2.5e-7, (k, adam[k], moved[k])
    for k in got[0][3]:
        assert np.array_equal(got[0][3][k].view(np.uint32), got[1][3][k].view(np.uint32)), k
    assert np.array_equal(got[0][4], got[1][4]) and np.array_equal(got[0][5], got[1][5])
    assert got[0][5].max() == 2.0 and got[0][4].max() > 0


def _refine_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import brush_amd
        from brush_amd import dist as BD

        dev = torch.device("cuda:0")
        n, w, h, deg = 6000, 160, 96, 1
        ncoef = (deg + 1) ** 2
        cloud = H.synthetic_cloud(n, deg, seed=8, mean_mult=0.002)
        splats = brush_amd.Splats(*(torch.from_numpy(cloud[k]).to(dev) for k in ("means", "sh", "quats", "raw_opac",
                                                                                  "log_scales")))
        c = H.reference_test_camera(w, h)
        cam = brush_amd.Camera([0.4 * rank, -0.2 * rank, -8.0], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
        torch.manual_seed(5 + rank)
        gt = torch.rand((h, w, 3), device=dev)
        # refinement every 3 steps with a threshold low enough that it clones / splits / prunes at once
        cfg = brush_amd.TrainConfig(warmup_steps=0, refine_every=3, densify_grad_thresh=1e-7, max_refine_step=100)
        trainer = brush_amd.SplatTrainer(splats, cfg)
        xchg = BD.ViewExchange(n, ncoef, dev, packed=True)  # the exact-size form (one broadcast per view)
        counts, refines = [], []
        for _ in range(8):
            trainer.step(splats, cam, gt, 1.0, world, None, xchg)
            counts.append(splats.num_splats())
            if trainer.last_refine is not None:
                r = trainer.last_refine
                refines.append((r.num_split, r.num_cloned, r.num_transparent_pruned, r.num_scale_pruned))
        torch.cuda.synchronize()
        q.put((rank, counts, refines, xchg.n,
               {k: getattr(splats, k).detach().cpu().numpy() for k in ("means", "sh_coeffs", "rotation", "raw_opacity",
                                                                       "log_scales")},
               trainer.moment1.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_exchange_survives_refinement_two_ranks_one_gpu():
    """The splat count changes under a running ViewExchange (refine_splats clones / splits / prunes, train.rs:395-579):
    the reduction takes its count, its moment offsets and its [view][n] index from the PARAMETERS of each call, so the
    steps after a refinement update every splat of the new cloud, and both ranks still hold the same bits."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_refine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=500)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    counts, refines, xn, params, m1 = got[0]
    assert len(refines) >= 2 and any(sum(r) > 0 for r in refines), refines
    assert len(set(counts)) > 1, counts              # the cloud really changed size ...
    assert xn == counts[-1] == params["means"].shape[0]  # ... and the exchange followed it
    assert got[1][0] == counts and got[1][1] == refines
    for k in params:
        assert np.array_equal(params[k].view(np.uint32), got[1][3][k].view(np.uint32)), k
        assert np.isfinite(params[k]).all()
    assert np.array_equal(m1.view(np.uint32), got[1][4].view(np.uint32))
    assert m1.shape[0] == counts[-1] * (11 + 3 * 4)  # moments laid out for the refined cloud


def _deferred_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import brush_amd
        from brush_amd import dist as BD
        from brush_amd import render as R

        R.DETERMINISTIC = True  # bitwise reproducible gradients: the two optimizers can be compared bit for bit
        dev = torch.device("cuda:0")
        n, w, h, deg = 6000, 160, 96, 3
        ncoef = (deg + 1) ** 2
        cloud = H.synthetic_cloud(n, deg, seed=8, mean_mult=0.0003)
        cloud["log_scales"] = cloud["log_scales"] - 3.5
        mk = lambda: brush_amd.Splats(*(torch.from_numpy(cloud[k]).to(dev) for k in ("means", "sh", "quats", "raw_opac",
                                                                                       "log_scales")))
        a, b = mk(), mk()
        ta = brush_amd.SplatTrainer(a, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=True))
        tb = brush_amd.SplatTrainer(b, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=False))
        xa, xb = BD.ViewExchange(n, ncoef, dev), BD.ViewExchange(n, ncoef, dev)
        torch.manual_seed(5 + rank)
        gt = torch.rand((h, w, 3), device=dev)
        losses, lagged = [], 0
        for i in range(7):
            ang = 2.0 * math.pi * ((2 * i + rank) % 5) / 5.0  # the two ranks' views move around the cloud
            cam = brush_amd.Camera([4.0 * math.sin(ang), 0.0, -4.0 * math.cos(ang)],
                                   [0.0, -math.sin(ang / 2.0), 0.0, math.cos(ang / 2.0)], 0.4, 0.3, (0.5, 0.5))
            la, _, _ = ta.step(a, cam, gt, 1.0, world, None, xa)
            lb, _, _ = tb.step(b, cam, gt, 1.0, world, None, xb)
            losses.append((float(la), float(lb)))
            if ta._lazy is not None:
                lagged = max(lagged, int((ta._lazy_bufs[0] < ta.opt_time).sum()))
        ta.sync(a)
        same = {k: bool(torch.equal(getattr(a, k).detach(), getattr(b, k).detach()))
                for k in ("means", "sh_coeffs", "rotation", "raw_opacity", "log_scales")}
        same["moment1"] = bool(torch.equal(ta.moment1, tb.moment1))
        same["moment2"] = bool(torch.equal(ta.moment2, tb.moment2))
        torch.cuda.synchronize()
        q.put((rank, losses, lagged, ta._lazy is not None, same, a.sh_coeffs.detach().cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_deferred_sh_adam_in_the_view_exchange_two_ranks_one_gpu():
    """Deferred Adam of the SH block inside the data-parallel reduction (brush_reduce_view_records_adam): the blocks of
    splats NO view of the batch saw stay pending.  Two ranks, moving views, deterministic mode: the deferred and the
    eager trainer give the same losses step by step and, after sync(), the same bits in every parameter and moment — on
    each rank and across the ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_deferred_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        item = q.get(timeout=500)
        got[item[0]] = item[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in range(world):
        losses, lagged, was_lazy, same, _ = got[rank]
        assert was_lazy and lagged > 0, (rank, was_lazy, lagged)
        assert all(la == lb for la, lb in losses), (rank, losses)
        assert all(same.values()), (rank, same)
    assert np.array_equal(got[0][4].view(np.uint32), got[1][4].view(np.uint32))
