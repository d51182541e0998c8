#!/usr/bin/env python3
"""bench.py — forward+backward splat rasterizer throughput on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU over RCCL).  One "step" = one forward+backward pass of the hot path over one view
per GPU (BASELINE.json metric: fwd+bwd ms/view & iters/s, 1 M splats @1080p), on the seeded
synthetic cloud of SURVEY §8(d) S1 (render_bench.rs:32-133 distribution, seed 4), inputs resident
in HBM before the timed region.  For N>1 each rank renders its own view of the replicated cloud
and the dense parameter-gradient block is all-reduced (RCCL) inside the step;
scaling is weak (per-GPU work fixed).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

VALU_PEAK_GINST = 614.4  # wave64 VALU instructions/s: 256 CUs * 4 SIMDs * 2.4 GHz / 4 cycles
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--splats", type=int, default=1 << 20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--mean-mult", type=float, default=1.0, help="1.0 = 'base', 0.25 = 'dense' (render_bench.rs:24)")
    ap.add_argument("--max-intersects", type=int, default=0, help="0 = reference default min(N*T, 128*65535)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--profile-steps", type=int, default=20, help="extra steps with stage events (untimed)")
    ap.add_argument("--train-steps", type=int, default=20,
                    help="extra (untimed for `value`) steps of the full training iteration: loss + Adam groups")
    ap.add_argument("--dense-allreduce", action="store_true",
                    help="N>1: all-reduce the dense 52+12C B/splat block instead of all-gathering the compact "
                         "60 B/visible-splat records (brush_amd/dist.py)")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue every launch from the host instead of replaying one captured hipGraph per step")
    return ap.parse_args()


def synthetic_cloud(n, sh_degree, seed=4, mean_mult=1.0):
    """Host-generated so that CPU and GPU see identical bits."""
    from brush_amd.synthetic import synthetic_cloud as gen

    return gen(n, sh_degree, seed=seed, mean_mult=mean_mult)


def view_camera(rank, w, h):
    """Rank 0 uses the reference bench camera (render_bench.rs:163-174); other ranks orbit it."""
    import brush_amd

    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    fov_x, fov_y = brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h)
    ang = 0.35 * rank
    # rotation about y by `ang`, camera placed 8 units behind the origin along its own -z
    rot = [0.0, math.sin(ang / 2), 0.0, math.cos(ang / 2)]
    pos = [-8.0 * math.sin(ang), 0.0, -8.0 * math.cos(ang)]
    return brush_amd.Camera(pos, rot, fov_x, fov_y, (0.5, 0.5))


def algorithmic_bytes(n, V, I, P, T, C, fwd_only=False):
    """B = N(92+12C) + V(356+12C) + I(132+16p) + 56P + 24T, p = passes of 8 bits."""
    p = max(1, math.ceil(max(1, math.ceil(math.log2(T + 1))) / 8))
    if fwd_only:
        return n * 40 + V * (212 + 12 * C) + I * (56 + 16 * p) + P * 20 + T * 16
    return n * (92 + 12 * C) + V * (356 + 12 * C) + I * (132 + 16 * p) + P * 56 + T * 24


# Algorithmic HBM bytes of ONE launch of each single-kernel stage.
def stage_bytes(stage, n, V, I, P, T, C):
    return {
        "rasterize": I * 40 + P * 20 + T * 8,              # isect gid + record gather, img + final_index
        "rasterize_bwd": I * (40 + 36) + P * 36 + T * 8,   # gather + 9 float atomics / (tile,splat); out,v_out,final
        "project_bwd": n * 4 + V * (40 + 36 + 4) + n * (52 + 12 * C),
        "project_visible": V * (4 + 40 + 12 * C + 4 + 36 + 4 + 4) + (n - V) * 4,
    }[stage]


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (torch.distributed.run on
    127.0.0.1) from this process, which has not touched the GPU, and exit with their code."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(torch.distributed.run --nproc-per-node {args.gpus}) or drop the launcher")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # Rehearsal hooks (not used by the driver): BRUSH_BENCH_DEVICE pins every rank to one device and
    # BRUSH_DIST_BACKEND=gloo replaces RCCL, so the N>1 code path can be exercised on a 1-GPU box.
    if "BRUSH_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BRUSH_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("BRUSH_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    n_gpus = world

    import brush_amd
    from brush_amd import dist as BD
    from brush_amd import render as R
    from brush_amd.profiler import StageProfiler

    n, w, h, deg = args.splats, args.width, args.height, args.sh_degree
    C = (deg + 1) ** 2
    cloud = synthetic_cloud(n, deg, mean_mult=args.mean_mult)
    p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
    cam = view_camera(rank, w, h)
    cap = args.max_intersects or None
    # upstream gradient of mean(img) (render_bench.rs:180)
    v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), dtype=torch.float32, device=dev)
    layout, total = R.grad_block_layout(n, C)
    block = torch.zeros(total, dtype=torch.float32, device=dev)

    def fwd_bwd():
        out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"],
                                      False, cap)
        R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out, block)
        return aux

    # The op never syncs, allocates or reads a count back, so one fwd+bwd is a capturable launch
    # sequence: replaying it as a hipGraph removes the ~3.5 us/launch host enqueue cost
    # (DESIGN.md "launch floor").  The all-reduce stays outside the graph.
    graph = None
    if not args.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd_bwd()  # warm allocator + code objects before capture
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            graph_aux = fwd_bwd()

    def step():
        if graph is not None:
            graph.replay()
            aux = graph_aux
        else:
            aux = fwd_bwd()
        if world > 1:  # sum of the per-view gradients on every rank (RCCL over xGMI)
            if args.dense_allreduce:
                BD.allreduce_param_grads(block, n, C)
            else:
                BD.allreduce_param_grads_compact(block, aux, p["means"], n, C)
        return aux

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        aux = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        aux = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * args.steps / elapsed  # whole-job views/s

    V, I = aux.read_num_visible(), aux.read_num_intersections()
    overflow = int(aux.overflow.item())
    P, T = w * h, (-(-w // 16)) * (-(-h // 16))

    # ---- per-stage device time (hipEvents on the op's stream), separate untimed steps ----
    stage_ms = {}
    with StageProfiler() as prof:
        acc = None
        for _ in range(max(1, args.profile_steps)):
            fwd_bwd()  # eager: events cannot be recorded inside a replayed graph
            ms = prof.read_ms()
            acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
        stage_ms = {k: v / max(1, args.profile_steps) for k, v in acc.items()}
    dominant = max(("rasterize", "rasterize_bwd", "project_bwd", "project_visible"), key=lambda k: stage_ms[k])
    dom_bytes = stage_bytes(dominant, n, V, I, P, T, C)
    achieved = dom_bytes / (stage_ms[dominant] * 1e-3) / 1e9 if stage_ms[dominant] > 0 else 0.0
    traffic, valu = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dominant)
            insts = tj.get("valu_insts", {}).get(dominant)
            if insts and stage_ms[dominant] > 0:
                # The ceiling that actually binds the compositing kernels: one non-packed
                # wave64 VALU instruction occupies a SIMD for 4 cycles -> 1024 SIMDs * 2.4 GHz / 4.
                rate = insts / (stage_ms[dominant] * 1e-3) / 1e9
                valu = {"insts_per_launch": int(insts), "achieved_Ginst_s": round(rate, 1), "peak_Ginst_s": VALU_PEAK_GINST,
                        "frac": round(rate / VALU_PEAK_GINST, 4),
                        "note": "SQ_INSTS_VALU per launch from profiles/ (rocprofv3 --pmc) / live kernel time"}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes": int(dom_bytes), "kernel_ms": round(stage_ms[dominant], 5), "valu_issue": valu}
    # Device-to-device copy bandwidth measured in the same run: 1 GiB read + 1 GiB write.
    src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    B = algorithmic_bytes(n, V, I, P, T, C)
    whole_path = {"algorithmic_bytes": int(B), "achieved_GBs": round(B / (ms_per_step * 1e-3) / 1e9, 2),
                  "frac_of_hbm_peak": round(B / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                  "measured_copy_GBs": round(copy_gbs, 1)}

    # ---- full training iteration: render + L1/SSIM loss + backward + 5 Adam groups
    train = None
    if args.train_steps > 0 and world == 1:  # secondary figure, single GPU only
        splats = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
        trainer = brush_amd.SplatTrainer(splats, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0))
        gt = torch.rand((h, w, 3), dtype=torch.float32, device=dev)  # synthetic target image

        for _ in range(3):
            trainer.step(splats, cam, gt, 1.0, 1, None)
        barrier()
        tt = time.perf_counter()
        for _ in range(args.train_steps):
            trainer.step(splats, cam, gt, 1.0, 1, None)
        barrier()
        tsec = time.perf_counter() - tt
        if world > 1:
            t = torch.tensor([tsec], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tsec = float(t.item())
        train = {"iters_per_s": round(args.train_steps / tsec, 2), "ms_per_iter": round(tsec * 1e3 / args.train_steps, 4),
                 "views_per_iter": n_gpus, "steps": args.train_steps,
                 "what": "render + L1*0.8-SSIM*0.2 loss + backward + 5 Adam groups (train.rs:211-359), no refinement, "
                         "fused HIP loss kernels around the op, Adam inside the backward's last kernel (brush_l1_ssim_loss, "
                         "brush_render_backward_adam)"}
        del splats, trainer, gt

    # ---- CPU baseline: the oracle (a port), rank 0, N=1 only --------------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O

        u_np = R.uniforms_to_numpy(aux)
        v_np = v_out.cpu().numpy()
        reps = max(1, args.cpu_reps)
        tc = time.perf_counter()
        for _ in range(reps):
            o_out, o_aux = O.render_forward(u_np, cloud["means"], cloud["log_scales"], cloud["quats"], cloud["sh"],
                                            cloud["raw_opac"], max_intersects=aux.max_intersects)
            O.render_backward(u_np, o_aux, cloud["means"], cloud["log_scales"], cloud["quats"], cloud["raw_opac"],
                              o_out, v_np)
        cpu_s = (time.perf_counter() - tc) / reps
        cpu_baseline = {"value": round(1.0 / cpu_s, 4), "unit": "views/s", "cores": O.num_threads(), "kind": "port",
                        "sample": f"{reps} fwd+bwd passes of the same workload (CPU restatement of the reference "
                                  f"algorithm, OpenMP; not wgpu/lavapipe), {cpu_s * 1e3:.0f} ms/view"}

    if rank == 0:
        line = {
            "metric": "fwd+bwd views/s (train-iter rate of the rasterizer path), 1M splats @1080p",
            "value": round(value, 3), "unit": "views/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'S1' if (n, w, h, deg, args.mean_mult) == (1 << 20, 1920, 1080, 3, 1.0) else 'custom'}: {n} splats @{w}x{h}, SH degree {deg}, seed 4, mean_mult {args.mean_mult}, "
                                   f"fwd+bwd per view", "views_per_step": n_gpus,
                       "parallelism": f"view-sharded dp{n_gpus}" if n_gpus > 1 else "single GPU",
                       "gradient_exchange": None if n_gpus == 1 else (
                           "dense all-reduce" if args.dense_allreduce else
                           "all-gather of compact per-view records + local expansion (same dense sum)"),
                       "launch": "eager" if graph is None else "hipGraph replay of one fwd+bwd",
                       "num_visible": V, "num_intersections": I, "max_intersects": aux.max_intersects,
                       "overflow": overflow},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "whole_path": whole_path, "train": train,
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
