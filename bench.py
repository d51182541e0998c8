#!/usr/bin/env python3
"""bench.py — forward+backward splat rasterizer throughput on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1 under torch.distributed.run, one
rank per GPU over RCCL).  One "step" = one forward+backward pass of the hot path over one view
per GPU (BASELINE.json metric: fwd+bwd ms/view & iters/s, 1 M splats @1080p), on the seeded
synthetic cloud of SURVEY §8(d) S1 (render_bench.rs:32-133 distribution, seed 4), inputs resident
in HBM before the timed region.  For N>1 each rank renders its own view of the replicated cloud
and the step ends with the SUM of the per-view parameter gradients on every rank: by
default through an RCCL all-gather of 64-byte per-visible-splat records and a deterministic
per-splat reduction (brush_amd/dist.py), with --dense-allreduce through one all-reduce of the dense
block; scaling is weak (per-GPU work fixed).  Rank 0 prints ONE JSON line; `train` in it is the full
training iteration (loss + Adam) at the same N.  At N=1 `value` is ONE fixed launch mode: the K steps as eager launches
through the C ABI after W warm-up steps — how the reference's host drives its kernels, and the mode BENCH_r03's value was
in (0.3475 ms; r01 / r02 quoted a hipGraph replay).  The same K steps replayed as one captured hipGraph are timed beside
it (`whole_path.graph_ms_per_step`; r02 0.374, r03 0.354-0.356 ms), so both series can be followed round to round.  `stage_ms` are IN-SITU stage times: the captured prefixes
cull .. stage k of the step are timed as replayed graphs and differenced (brush_profiler_stop_after), so they add up to
the step; `stage_ms_events` are the hipEvent times of eager passes that earlier rounds reported (every event record adds
~3-5 us to its stage).  At N>1 the three gradient-exchange forms (records padded / records packed / dense all-reduce)
are timed in the same run (`exchange_variants`) and `value` is the fastest.
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

# the launch-bound middle of the forward + the accumulator zero-fill (VERDICT r03 "seven small stages")
SMALL_STAGES = ("depth_sort", "project_visible", "prefix_sum", "map_intersects", "tile_sort", "tile_bins", "bwd_zero")
VALU_PEAK_GINST = 923.9  # wave64 v_fma_f32/s: 256 CUs * 4 SIMDs * 2.4 GHz / 2.66 cycles (measured, tools/ubench)
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
                    help="N>1: all-reduce the dense 52+12C B/splat block instead of all-gathering the 64 B/visible-splat "
                         "records (brush_amd/dist.py)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra workloads (dense scene, S3) of the N=1 line")
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


def kernel_sources_sha16():
    """Fingerprint of the kernel sources (brush_amd/csrc/*.hip, *.hpp, include/brush_hip.h): profiles/traffic.json
    carries the one it was taken at (tools/summarize_profiles.py), so a counter figure older than the kernels is
    reported as stale instead of being passed off as this build's traffic."""
    import glob
    import hashlib

    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "brush_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "brush_amd", "csrc", "*.hpp")) +
                   [os.path.join(ROOT, "include", "brush_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(n, V, I, P, T, C, fwd_only=False):
    """B = N(92+12C) + V(356+12C) + I(132+16p) + 56P + 24T, p = passes of 8 bits."""
    p = max(1, math.ceil(max(1, math.ceil(math.log2(T + 1))) / 8))
    if fwd_only:
        return n * 40 + V * (212 + 12 * C) + I * (56 + 16 * p) + P * 20 + T * 16
    return n * (92 + 12 * C) + V * (356 + 12 * C) + I * (132 + 16 * p) + P * 56 + T * 24


# Algorithmic HBM bytes of ONE launch of each single-kernel stage.  Since round 4 the dense
# gradients' zeros (N (52 + 12 C) bytes, SURVEY §8d's "dense v_* written exactly once") are stored by the compositing
# backward in passing and the VJP kernel writes the visible splats' rows only: the bytes moved with the stage that
# writes them, the whole-path sum is unchanged.
def stage_bytes(stage, n, V, I, P, T, C):
    return {
        "rasterize": I * 40 + P * 20 + T * 8,              # isect gid + record gather, img + final_index
        # gather + 9 float atomics / (tile,splat); out, v_out, final_index; the zeros of the six dense gradient arrays
        "rasterize_bwd": I * (40 + 36) + P * 36 + T * 8 + (n - V) * (52 + 12 * C),
        "project_bwd": n * 4 + V * (40 + 36 + 4) + V * (52 + 12 * C),
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(sec):
        if world > 1:
            t = torch.tensor([sec], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return sec

    def timed(fn, steps, warmup):
        """W untimed + K timed calls bracketed by barrier + synchronize on both sides, max over ranks."""
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    class Workload:
        """One view of one cloud: forward (+ dense backward) as eager launches or as one replayed hipGraph."""

        def __init__(self, pp, nn, ww, hh, camera, capacity):
            self.p, self.n, self.w, self.h, self.cam, self.cap = pp, nn, ww, hh, camera, capacity
            self.C = pp["sh"].shape[1]
            # upstream gradient of mean(img) (render_bench.rs:180)
            self.v_out = torch.full((hh, ww, 4), 1.0 / (4 * ww * hh), dtype=torch.float32, device=dev)
            self.block = torch.zeros(R.grad_block_layout(nn, self.C)[1], dtype=torch.float32, device=dev)
            self.graph = self.graph_fwd = None

        def forward(self):
            q = self.p
            return R._forward_impl(self.cam, (self.w, self.h), q["means"], q["log_scales"], q["quats"], q["sh"],
                                   q["raw_opac"], False, self.cap)

        def fwd_bwd(self):
            q = self.p
            out, aux, u = self.forward()
            R._backward_impl(u, aux, q["means"], q["log_scales"], q["quats"], q["raw_opac"], self.C, out, self.v_out,
                             self.block)
            return aux

        def _capture(self, fn):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()  # warm allocator + code objects before capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                res = fn()
            return g, res

        def capture(self):
            # The op never syncs, allocates or reads a count back, so one fwd+bwd is a capturable launch sequence:
            # replaying it as a hipGraph removes the ~3.5 us/launch host enqueue cost (DESIGN.md "launch floor").
            self.graph, self.graph_aux = self._capture(self.fwd_bwd)

        def capture_forward(self):
            self.graph_fwd, self.fwd_res = self._capture(self.forward)

        def step(self):
            if self.graph is not None:
                self.graph.replay()
                return self.graph_aux
            return self.fwd_bwd()

    wl = Workload(p, n, w, h, cam, cap)
    exchange_mode = None
    exchange_variants = None
    graph_ms = None
    if world == 1:
        if not args.no_graph:  # the same K steps as one replayed hipGraph, reported beside the headline
            wl.capture()
            graph_ms = timed(wl.step, args.steps, args.warmup) * 1e3 / args.steps
        step = wl.fwd_bwd
        elapsed = timed(step, args.steps, args.warmup)
    else:
        # N > 1: one view per rank; the step ends with the SUM of the per-view parameter gradients on every rank.  Three
        # exchange forms, each W warm-up + K timed steps in this run; `value` is the fastest (the ranking on xGMI is not
        # known before the first hardware run: DESIGN.md §6).
        NAMES = {
            "records_padded": "all-gather (one all_gather_into_tensor, views padded to the largest) of 64-byte "
                              "per-visible-splat gradient records + deterministic per-splat sum over views into the dense "
                              "block (same sum on every rank, bit for bit)",
            "records_packed": "exact-size exchange (one broadcast per view) of the 64-byte records + the same "
                              "deterministic per-splat sum",
            "dense_allreduce": "dense all-reduce of the [v_means|v_scales|v_quats|v_opac|v_sh] block (52+12C B/splat)",
        }
        if not args.no_graph:
            wl.capture_forward()

        def make_records_step(xchg):
            def record_step():
                if wl.graph_fwd is not None:
                    wl.graph_fwd.replay()
                    out, aux, u = wl.fwd_res
                else:
                    out, aux, u = wl.forward()
                xchg.begin(aux)
                xchg.backward_records(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], out, wl.v_out)
                xchg.gather()
                xchg.reduce_dense(p["means"], wl.block)
                return aux
            return record_step

        def dense_step():
            aux = wl.fwd_bwd()  # eager: the forward graph above and a fwd+bwd graph would double the captured pools
            BD.allreduce_param_grads(wl.block, n, C)
            return aux

        xchgs = {"records_padded": BD.ViewExchange(n, C, dev, packed=False),
                 "records_packed": BD.ViewExchange(n, C, dev, packed=True)}
        steps_by_name = {"records_padded": make_records_step(xchgs["records_padded"]),
                         "records_packed": make_records_step(xchgs["records_packed"]), "dense_allreduce": dense_step}
        todo = ["dense_allreduce"] if args.dense_allreduce else list(steps_by_name)
        exchange_variants, best = {}, None
        for name in todo:
            sec = timed(steps_by_name[name], args.steps, args.warmup)
            exchange_variants[name] = {"ms_per_step": round(sec * 1e3 / args.steps, 4),
                                       "views_per_s": round(n_gpus * args.steps / sec, 3), "what": NAMES[name]}
            x = xchgs.get(name)
            if x is not None:  # did the host ever wait for the per-view counts (ViewExchange.counts)?
                exchange_variants[name]["host_waits"] = x.host_waits
                exchange_variants[name]["host_wait_ms_per_step"] = round(x.host_wait_s * 1e3 / (args.steps + args.warmup), 4)
            if best is None or sec < best[1]:
                best = (name, sec)
        step, elapsed = steps_by_name[best[0]], best[1]
        exchange_mode = NAMES[best[0]] + (" — fastest of the forms timed in this run" if len(todo) > 1 else "")
    aux = step()
    torch.cuda.synchronize()
    ms_per_step = elapsed * 1e3 / args.steps
    value = n_gpus * args.steps / elapsed  # whole-job views/s

    V, I = aux.read_num_visible(), aux.read_num_intersections()
    overflow = int(aux.overflow.item())
    P, T = w * h, (-(-w // 16)) * (-(-h // 16))

    eager_ms = ms_per_step if world == 1 else None
    launch_used = None

    # ---- per-stage device time (hipEvents on the op's stream), separate untimed steps ----
    def profile_stages(workload, steps):
        """Mean stage times (ms) of `steps` eager fwd+bwd passes with hipEvents recorded on the op's stream after every
        stage (brush_profiler_*): events cannot be recorded inside a replayed graph."""
        with StageProfiler() as prof:
            acc = None
            for _ in range(max(1, steps)):
                workload.fwd_bwd()
                ms = prof.read_ms()
                acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
        return {k: v / max(1, steps) for k, v in acc.items()}

    def dominant_roofline(st_ms, nn, VV, II, PP, TT, CC):
        """`roofline` object of the single-kernel stage that takes the longest: algorithmic bytes per launch (SURVEY
        §8d accounting, stage_bytes) / its live event time, against the HBM peak."""
        dom = max(("rasterize", "rasterize_bwd", "project_bwd", "project_visible"), key=lambda k: st_ms[k])
        nbytes = stage_bytes(dom, nn, VV, II, PP, TT, CC)
        ach = nbytes / (st_ms[dom] * 1e-3) / 1e9 if st_ms[dom] > 0 else 0.0
        return dom, nbytes, ach

    def profile_stages_in_situ(workload, replays=12, rounds=3, copies=4):
        """Stage times without event overhead: for every stage k the prefix cull .. k of the step is captured as a graph
        (`copies` back-to-back passes per graph, so a short prefix is not bound by the host's replay rate), replayed,
        and the prefix times are differenced.  The last prefix is the whole step."""
        res, prev = {}, 0.0
        with StageProfiler() as prof:
            names = prof.names
            n_fwd = names.index("rasterize") + 1
            try:
                for k, name in enumerate(names):
                    prof.stop_after(name)

                    def fn():
                        for _ in range(copies):
                            q = workload.p
                            out, aux, u = workload.forward()
                            if k >= n_fwd:
                                R._backward_impl(u, aux, q["means"], q["log_scales"], q["quats"], q["raw_opac"], workload.C,
                                                 out, workload.v_out, workload.block)

                    g, _ = workload._capture(fn)
                    best = None
                    for _ in range(rounds):
                        g.replay()
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for _ in range(replays):
                            g.replay()
                        torch.cuda.synchronize()
                        t = (time.perf_counter() - t0) * 1e3 / (replays * copies)
                        best = t if best is None else min(best, t)
                    res[name] = max(best - prev, 0.0)
                    prev = best
                    del g
            finally:
                prof.stop_after(None)
        torch.cuda.empty_cache()
        return res, prev

    stage_ms_events = profile_stages(wl, args.profile_steps)
    stage_ms, prefix_total_ms = (None, None)
    if world == 1 and args.profile_steps > 0:
        stage_ms, prefix_total_ms = profile_stages_in_situ(wl)
    if stage_ms is None:
        stage_ms = stage_ms_events
    # the roofline object prices the dominant kernel with its hipEvent time on the op's stream (the contract's clock)
    dominant, dom_bytes, achieved = dominant_roofline(stage_ms_events, n, V, I, P, T, C)
    traffic, valu, traffic_src, traffic_stale = None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tj = {}
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dominant)
            fresh = tj.get("_csrc_sha16") == kernel_sources_sha16()
            traffic_src = (f"profiles/traffic.json (tag {tj.get('_tag', 'static')}, taken at commit {tj.get('_commit', '?')}): "
                           "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload (tools/profile_gpu.sh), not "
                           "re-measured in this run; " +
                           ("the kernel sources are the ones the counters were taken on" if fresh else
                            "STALE: the kernel sources have changed since (fingerprint mismatch), so the figure is "
                            "reported as traffic_stale and `traffic` is null"))
            if not fresh:
                traffic_stale, traffic = traffic, None
            insts = tj.get("valu_insts", {}).get(dominant) if fresh else None
            if insts and stage_ms_events[dominant] > 0:
                # The ceiling that binds the compositing kernels: VALU issue.  Measured with
                # tools/ubench/valu_rate.hip: one v_fma_f32 per 2.66 SIMD cycles at 8 waves/SIMD.
                rate = insts / (stage_ms_events[dominant] * 1e-3) / 1e9
                valu = {"insts_per_launch": int(insts), "achieved_Ginst_s": round(rate, 1), "peak_Ginst_s": VALU_PEAK_GINST,
                        "frac": round(rate / VALU_PEAK_GINST, 4),
                        # every VALU lane-op counted as one FMA (2 flop): an upper bound of the fraction of the
                        # 157.3 TFLOP/s vector peak (SURVEY 8d asks for this figure)
                        "valu_flops_frac_upper_bound": round(insts * 64 * 2 / (stage_ms_events[dominant] * 1e-3) / 157.3e12, 4),
                        "note": "SQ_INSTS_VALU per launch from profiles/ (rocprofv3 --pmc) / live kernel time; peak = "
                                "1024 SIMDs * 2.4 GHz / 2.66 cycles per v_fma_f32 (measured, profiles/r02a_valu_rate.txt)"}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": int(dom_bytes), "kernel_ms": round(stage_ms_events[dominant], 5),
                "kernel_ms_in_situ": round(stage_ms[dominant], 5), "valu_issue": valu}
    if traffic is not None and stage_ms_events[dominant] > 0:
        # what the memory system really moved for this kernel, against the same peak
        roofline["frac_traffic"] = round(traffic / (stage_ms_events[dominant] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if traffic_stale is not None:
        roofline["traffic_stale"] = traffic_stale
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
                  "measured_copy_GBs": round(copy_gbs, 1),
                  "eager_ms_per_step": None if eager_ms is None else round(eager_ms, 4),
                  "graph_ms_per_step": None if graph_ms is None else round(graph_ms, 4)}

    # ---- full training iteration: render + L1/SSIM loss + backward + 5 Adam groups.
    # N>1: one view per rank, records exchanged, every rank applies the same Adam update (train.rs:229-359 for a
    # batch of N views); iters/s counts optimizer steps, views/s = N * iters/s.
    train = None
    if args.train_steps > 0:
        splats = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
        trainer = brush_amd.SplatTrainer(splats, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0))
        torch.manual_seed(1234 + rank)
        gt = torch.rand((h, w, 3), dtype=torch.float32, device=dev)  # synthetic target image
        txchg = BD.ViewExchange(n, C, dev) if world > 1 else None

        def train_step():
            trainer.step(splats, cam, gt, 1.0, world, None, txchg)

        def timed_train(tr, sp, step_fn, steps):
            """K iterations AND the flush of whatever optimizer steps they deferred (SplatTrainer.sync: the SH blocks
            of splats the views never saw), inside the timed region: no optimizer work is left out of the clock."""
            for _ in range(3):
                step_fn()
            tr.sync(sp)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step_fn()
            tr.sync(sp)
            barrier()
            return max_over_ranks(time.perf_counter() - t0)

        tsec = timed_train(trainer, splats, train_step, args.train_steps)
        train = {"iters_per_s": round(args.train_steps / tsec, 2), "ms_per_iter": round(tsec * 1e3 / args.train_steps, 4),
                 "views_per_iter": n_gpus, "views_per_s": round(n_gpus * args.train_steps / tsec, 2), "steps": args.train_steps,
                 "what": "render + L1*0.8-SSIM*0.2 loss + backward + 5 Adam groups (train.rs:211-359), no refinement, "
                         "fused HIP loss kernels around the op; " + (
                             "Adam inside the backward (brush_render_backward_adam) with the SH block's zero-gradient "
                             "steps deferred and replayed bit-exactly (BrushLazySh); the final flush of the K iterations "
                             "is inside the timed region" if world == 1 else
                             "per-view gradient records all-gathered (RCCL), summed per splat and fed straight into Adam "
                             "(brush_reduce_view_records_adam), densification statistics inside the records; the SH "
                             "blocks of splats no view of the batch saw stay pending (BrushLazySh), the final flush is "
                             "inside the timed region")}
        if world == 1:
            # the same K iterations with every Adam step applied when it happens (the reference's schedule), and both
            # optimizers on a cycle of 8 different views (view_camera), where deferred blocks do get caught up by the
            # forward and the backward of later views
            def run(deferred, cams):
                sp = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
                tr = brush_amd.SplatTrainer(sp, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0,
                                                                      deferred_sh_adam=deferred))
                i = [0]

                def step_fn():
                    tr.step(sp, cams[i[0] % len(cams)], gt, 1.0, 1, None, None)
                    i[0] += 1

                sec = timed_train(tr, sp, step_fn, args.train_steps)
                del sp, tr
                return round(sec * 1e3 / args.train_steps, 4)

            train["eager_adam_ms_per_iter"] = run(False, [cam])
            ring = [view_camera(v, w, h) for v in range(8)]
            train["changing_views"] = {"views": 8, "ms_per_iter": run(True, ring), "eager_adam_ms_per_iter": run(False, ring),
                                       "what": "the same iteration over a cycle of the 8 views of the multi-GPU bench"}
        if world > 1:
            # replicated parameters must stay replicated: compare a checksum of the updated parameters across ranks
            chk = torch.stack([splats.means.detach().double().sum(), splats.sh_coeffs.detach().double().sum(),
                               splats.rotation.detach().double().sum()])
            lo, hi = chk.clone(), chk.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            train["parameters_identical_on_all_ranks"] = bool(torch.equal(lo, hi))
        del splats, trainer, gt

    # ---- other named workloads (N=1): the dense scene and S3, short runs, reported as extra keys -------------
    extra = None
    if world == 1 and not args.no_extra:
        extra = {}
        todo = [("dense_scene", dict(n=n, w=w, h=h, deg=deg, mean_mult=0.25, cap=None, steps=20)),
                # BASELINE config 3 ("full sort/composite stress"): 3 M splats packed until the frame saturates,
                # 27 M intersections = 3.2x the reference's cap, so the capacity is raised (tests: test_c3_scale_raised_cap)
                ("c3", dict(n=3_000_000, w=w, h=h, deg=deg, mean_mult=0.12, cap=40_000_000, steps=6)),
                ("S3", dict(n=20_971_520, w=3840, h=2160, deg=deg, mean_mult=1.0, cap=24_000_000, steps=6))]
        if (n, w, h, args.mean_mult) != (1 << 20, 1920, 1080, 1.0):
            todo = []
        for name, c in todo:
            try:
                cl = synthetic_cloud(c["n"], c["deg"], mean_mult=c["mean_mult"])
                pp = {k: torch.as_tensor(v, device=dev) for k, v in cl.items()}
                del cl
                x = Workload(pp, c["n"], c["w"], c["h"], view_camera(0, c["w"], c["h"]), c["cap"])
                sec = timed(x.fwd_bwd, c["steps"], 2)  # eager launches, like the headline
                ax = x.fwd_bwd()
                torch.cuda.synchronize()
                Vx, Ix = ax.read_num_visible(), ax.read_num_intersections()
                Px, Tx = c["w"] * c["h"], (-(-c["w"] // 16)) * (-(-c["h"] // 16))
                Cx = (c["deg"] + 1) ** 2
                Bx = algorithmic_bytes(c["n"], Vx, Ix, Px, Tx, Cx)
                st = profile_stages(x, 3)
                dom, dbytes, ach = dominant_roofline(st, c["n"], Vx, Ix, Px, Tx, Cx)
                tr = tj.get("extra", {}).get(name, {}).get(dom) if tj.get("_csrc_sha16") == kernel_sources_sha16() else None
                extra[name] = {"workload": f"{c['n']} splats @{c['w']}x{c['h']}, SH degree {c['deg']}, mean_mult {c['mean_mult']}",
                               "ms_per_step": round(sec * 1e3 / c["steps"], 4), "num_visible": Vx, "num_intersections": Ix,
                               "max_intersects": ax.max_intersects, "overflow": int(ax.overflow.item()),
                               "algorithmic_bytes": int(Bx),
                               "frac_of_hbm_peak": round(Bx / (sec / c["steps"]) / 1e9 / HBM_PEAK_GBS, 5),
                               "stage_ms": {k: round(v, 5) for k, v in st.items()},
                               "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
                                            "algorithmic_bytes": int(dbytes), "kernel_ms": round(st[dom], 5),
                                            "traffic": tr,
                                            # counter bytes / kernel time: the fraction that means something when the
                                            # lists are not walked to their end (c3: the saturated frame stops early,
                                            # the I*76 formula credits bytes that never move)
                                            "frac_traffic": None if not tr or st[dom] <= 0 else round(
                                                tr / (st[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}}
                del x, pp, ax
                torch.cuda.empty_cache()
            except Exception as e:  # an extra line must never take the headline down
                extra[name] = {"error": repr(e)}

    # ---- CPU baseline: the oracle (a port), rank 0, N=1 only --------------------------------
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O

        u_np = R.uniforms_to_numpy(aux)
        v_np = wl.v_out.cpu().numpy()
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
        launch = launch_used if launch_used else (
            "eager launches through the C ABI (hipGraph replay of the same step: whole_path.graph_ms_per_step)" if world == 1 else
            "record forms: hipGraph replay of the forward, then backward, exchange and reduction enqueued per step; dense "
            "all-reduce form: eager launches")
        line = {
            "metric": "fwd+bwd views/s (train-iter rate of the rasterizer path), 1M splats @1080p",
            "value": round(value, 3), "unit": "views/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{'S1' if (n, w, h, deg, args.mean_mult) == (1 << 20, 1920, 1080, 3, 1.0) else 'custom'}: {n} splats @{w}x{h}, SH degree {deg}, seed 4, mean_mult {args.mean_mult}, "
                                   f"fwd+bwd per view", "views_per_step": n_gpus,
                       "parallelism": f"view-sharded dp{n_gpus}" if n_gpus > 1 else "single GPU",
                       "gradient_exchange": exchange_mode, "launch": launch,
                       "num_visible": V, "num_intersections": I, "max_intersects": aux.max_intersects,
                       "overflow": overflow},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "whole_path": whole_path, "train": train,
            "stage_ms": {k: round(v, 5) for k, v in stage_ms.items()},
            "stage_ms_method": ("in situ: differences of replayed prefix graphs cull..stage (brush_profiler_stop_after); sum = "
                                f"{sum(stage_ms.values()):.4f} ms vs the step's {ms_per_step:.4f}") if prefix_total_ms else
                               "hipEvents between the stages of eager passes",
            "stage_ms_events": {k: round(v, 5) for k, v in stage_ms_events.items()},
            "small_stages_ms": {"in_situ": round(sum(stage_ms[k] for k in SMALL_STAGES), 5),
                                "events": round(sum(stage_ms_events[k] for k in SMALL_STAGES), 5),
                                "stages": list(SMALL_STAGES)},
            "exchange_variants": exchange_variants, "extra_workloads": extra,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
