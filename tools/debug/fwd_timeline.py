"""Development aid: where the microseconds of every kernel of one fwd+bwd go (needs `make -C brush_amd/csrc trace`,
which builds brush_amd/csrc/build/libbrush_hip_trace.so with the s_memrealtime stamps of trace.hpp).

    python tools/debug/fwd_timeline.py [S1|dense] [out.json]

One replay of the captured fwd+bwd graph is traced.  Per launch, over the workgroups that did work (wave 0 of each):
first / last entry, the segments between the marks (medians and maxima), last exit, and the gap to the previous launch's
last exit.  The stamps are 10 ns ticks of one chip-wide counter.  The trace build is slower than the product (every
mark waits for the value it names): read shares and segment lengths, not the total."""
import ctypes
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["BRUSH_HIP_LIB"] = os.path.join(ROOT, "brush_amd", "csrc", "build", "libbrush_hip_trace.so")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import brush_amd  # noqa: E402
from brush_amd import _lib  # noqa: E402
from brush_amd import render as R  # noqa: E402
from brush_amd.synthetic import synthetic_cloud  # noqa: E402

CFG = {"S1": (1 << 20, 1920, 1080, 3, 1.0, None), "dense": (1 << 20, 1920, 1080, 3, 0.25, None),
       "c3": (3_000_000, 1920, 1080, 3, 0.12, 40_000_000), "S3": (20_971_520, 3840, 2160, 3, 1.0, 24_000_000)}
KNAMES = {1: "k_project_cull", 2: "k_compact", 3: "k_sort_upsweep", 4: "k_sort_downsweep", 5: "k_project_visible",
          6: "k_walk_count", 7: "k_scan_reduce", 8: "k_scan_down", 9: "k_map_intersects", 10: "k_rasterize_quad",
          11: "k_zero_compact_grads", 12: "k_rasterize_backward_quad", 13: "k_project_backward", 14: "k_sort_scan",
          15: "k_sort_downsweep_big", 16: "k_sort_onesweep"}
# what marks 1..4 mean per kernel (segment i = mark i - previous taken stamp)
MARKS = {15: ["*d_n arrived", "reordered in LDS (values loaded)", "keys arrived (barrier A)", "ranking + scans done (barrier C)"],
         1: ["means[0] arrived", "all phase-A loads arrived", "phase A done", "phase B done"],
         2: ["block counts summed", "keys arrived", "", ""],
         3: ["*d_n arrived", "first key arrived", "histogram done (2nd barrier)", ""],
         4: ["*d_n arrived", "count table summed", "barrier A + keys arrived", "ranking + scans done (barrier C)"],
         5: ["*num_visible arrived", "record gather arrived", "queue slots reserved", "inline walk done"],
         6: ["*counter arrived", "items + records arrived", "walk done", ""],
         7: ["*valid_n arrived", "tile loaded + summed", "", ""],
         8: ["tile sums before summed", "*valid_n arrived", "tile loaded", "scanned (before store)"],
         9: ["", "", "", ""], 10: ["", "", "", ""], 11: ["*num_visible arrived", "", "", ""], 12: ["", "", "", ""],
         13: ["", "", "", ""], 16: ["ticket", "keys arrived + published", "ranked", "look-back done"]}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "S1"
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    n, w, h, deg, mm, cap = CFG[name]
    dev = torch.device("cuda:0")
    p = {k: torch.as_tensor(v, device=dev) for k, v in synthetic_cloud(n, deg, seed=4, mean_mult=mm).items()}
    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w),
                           brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
    C = (deg + 1) ** 2
    v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), device=dev)
    block = torch.zeros(R.grad_block_layout(n, C)[1], device=dev)

    def fwd_bwd():
        out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"],
                                      False, cap)
        R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out, block)
        return aux

    L = _lib.lib()
    L.brush_debug_trace_begin.restype = ctypes.c_int
    L.brush_debug_trace_read.restype = ctypes.c_long
    L.brush_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.brush_debug_trace_begin() == 0
    for _ in range(3):
        aux = fwd_bwd()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    assert L.brush_debug_trace_begin() == 0
    g.replay()
    torch.cuda.synchronize()
    cap_rec = 20 * 8 * 8192  # trace.hpp: kTraceKids * kTraceLaunchSlots * kTraceBlocks
    buf = np.zeros((cap_rec, 8), dtype=np.uint64)
    got = L.brush_debug_trace_read(buf.ctypes.data_as(ctypes.c_void_p), cap_rec)
    assert got > 0, got
    a = buf[:got]
    a = a[a[:, 0] > 0]  # slots nobody wrote
    got = len(a)
    if out_path:
        np.save(os.path.splitext(out_path)[0] + "_raw.npy", a)
    t = a[:, :6].astype(np.int64)
    kid = (a[:, 6] & 0xFFFFFFFF).astype(np.int64)
    aux_w = (a[:, 6] >> 32).astype(np.int64)
    blk = (a[:, 7] & 0xFFFFFFFF).astype(np.int64)
    grid = (a[:, 7] >> 32).astype(np.int64)
    t_origin = t[:, 0].min()
    us = lambda x: (x - t_origin) / 100.0  # noqa: E731
    # a launch = records of one (kid, aux, grid) whose entries are closer than 30 us to each other, in time order
    order = np.argsort(t[:, 0], kind="stable")
    launches = []
    open_l = {}
    for i in order:
        key = (int(kid[i]), int(aux_w[i]), int(grid[i]))
        cur = open_l.get(key)
        if cur is None or t[i, 0] - cur["last_entry"] > 3000:
            cur = {"key": key, "idx": [], "last_entry": t[i, 0]}
            open_l[key] = cur
            launches.append(cur)
        cur["idx"].append(i)
        cur["last_entry"] = max(cur["last_entry"], t[i, 0])
    rows = []
    prev_exit = None
    for l in launches:
        idx = np.array(l["idx"])
        k, ax, gr = l["key"]
        tt = t[idx]
        worked = tt[:, 1:5].max(axis=1) > 0 if k not in (9, 10, 12, 13) else np.ones(len(idx), bool)
        ww = tt[worked] if worked.any() else tt
        row = {"kernel": KNAMES.get(k, str(k)), "aux": ax, "grid": gr, "workgroups_seen": int(len(idx)),
               "workgroups_with_work": int(worked.sum()),
               "first_entry_us": round(float(us(tt[:, 0].min())), 2), "last_entry_us": round(float(us(tt[:, 0].max())), 2),
               "last_exit_us": round(float(us(tt[:, 5].max())), 2),
               "duration_us": round(float((tt[:, 5].max() - tt[:, 0].min()) / 100.0), 2),
               "gap_from_prev_exit_us": None if prev_exit is None else round(float((tt[:, 0].min() - prev_exit) / 100.0), 2)}
        segs = []
        prev = ww[:, 0]
        for m in range(1, 5):
            taken = ww[:, m] > 0
            if not taken.any():
                continue
            d = (ww[taken, m] - prev[taken]) / 100.0
            segs.append({"until": MARKS.get(k, [""] * 4)[m - 1] or f"mark {m}", "median_us": round(float(np.median(d)), 2),
                         "max_us": round(float(d.max()), 2)})
            prev = np.where(taken, ww[:, m], prev)
        d = (ww[:, 5] - prev) / 100.0
        segs.append({"until": "exit (stores issued)", "median_us": round(float(np.median(d)), 2), "max_us": round(float(d.max()), 2)})
        row["segments"] = segs
        row["wave_life_median_us"] = round(float(np.median((ww[:, 5] - ww[:, 0]) / 100.0)), 2)
        rows.append(row)
        prev_exit = tt[:, 5].max()
    res = {"workload": name, "num_visible": aux.read_num_visible(), "num_intersections": aux.read_num_intersections(),
           "records": int(got), "launches": rows,
           "note": "trace build (libbrush_hip_trace.so): every mark waits for the value it names, so overlaps of the product "
                   "build are partly serialised; one replay of the captured fwd+bwd graph; times in us from the first entry"}
    for r in rows:
        seg = " | ".join(f"{s['until']}: {s['median_us']}/{s['max_us']}" for s in r["segments"])
        print(f"{r['kernel']:28s} aux={r['aux']:<3d} grid={r['grid']:<5d} work={r['workgroups_with_work']:<5d} "
              f"gap={r['gap_from_prev_exit_us']} entry={r['first_entry_us']}..{r['last_entry_us']} exit={r['last_exit_us']} "
              f"dur={r['duration_us']}  [{seg}]")
    if out_path:
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
