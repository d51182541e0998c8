"""Development aid: per-wave timeline of the compositing backward (needs `make -C brush_amd/csrc trace`:
libbrush_hip_trace.so, built with BRUSH_BWD_TRACE)."""
import ctypes, math, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["BRUSH_HIP_LIB"] = os.path.join(ROOT, "brush_amd", "csrc", "build", "libbrush_hip_trace.so")
import numpy as np
import torch
import brush_amd
from brush_amd import render as R, _lib
from brush_amd.synthetic import synthetic_cloud

CFG = {"S1": (1 << 20, 1920, 1080, 3, 1.0, None), "dense": (1 << 20, 1920, 1080, 3, 0.25, None)}
name = sys.argv[1] if len(sys.argv) > 1 else "S1"
n, w, h, deg, mm, cap = CFG[name]
dev = torch.device("cuda:0")
p = {k: torch.as_tensor(v, device=dev) for k, v in synthetic_cloud(n, deg, seed=4, mean_mult=mm).items()}
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
C = (deg + 1) ** 2
v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), device=dev)
block = torch.zeros(R.grad_block_layout(n, C)[1], device=dev)
for _ in range(4):
    out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, cap)
    R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out, block)
torch.cuda.synchronize()
L = _lib.lib()
buf = (ctypes.c_uint64 * (16384 * 4))()
L.brush_debug_read_bwd_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = L.brush_debug_read_bwd_trace(buf, 16384 * 4)
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0).astype(np.float64) / 100.0  # wall_clock64: 100 MHz -> us
end = (a[:, 1] - t0).astype(np.float64) / 100.0
hw = a[:, 2] & 0xFFFFFFFF
xcc = a[:, 2] >> 32
tiles = a[:, 3] >> 48
recs = (a[:, 3] >> 24) & 0xFFFFFF
res = {"workload": name, "rc": rc, "waves": int(len(a)), "kernel_us": float(end.max()),
       "start_us_pct": [float(np.percentile(start, q)) for q in (0, 25, 50, 75, 90, 100)],
       "end_us_pct": [float(np.percentile(end, q)) for q in (0, 5, 25, 50, 75, 95, 100)],
       "life_us_pct": [float(np.percentile(end - start, q)) for q in (0, 5, 50, 95, 100)],
       "tiles_per_wave": [int(tiles.min()), float(tiles.mean()), int(tiles.max())],
       "recs_per_wave_pct": [float(np.percentile(recs, q)) for q in (0, 5, 50, 95, 100)]}
# resident waves over time
ts = np.linspace(0, end.max(), 41)
res["resident_waves"] = [int(((start <= t) & (end > t)).sum()) for t in ts]
# per XCC finish
res["xcc_end_us"] = {int(x): float(end[xcc == x].max()) for x in np.unique(xcc)}
res["xcc_waves"] = {int(x): int((xcc == x).sum()) for x in np.unique(xcc)}
# per (xcc, cu, simd) grouping: HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
key = (xcc.astype(np.int64) << 16) | (hw & 0xFFF0).astype(np.int64)
uniq, inv = np.unique(key, return_inverse=True)
per_simd_end = np.array([end[inv == i].max() for i in range(len(uniq))])
per_simd_cnt = np.array([(inv == i).sum() for i in range(len(uniq))])
per_simd_recs = np.array([recs[inv == i].sum() for i in range(len(uniq))])
res["simds_seen"] = int(len(uniq))
res["simd_end_us_pct"] = [float(np.percentile(per_simd_end, q)) for q in (0, 5, 50, 95, 100)]
res["simd_waves_pct"] = [float(np.percentile(per_simd_cnt, q)) for q in (0, 5, 50, 95, 100)]
res["simd_recs_pct"] = [float(np.percentile(per_simd_recs, q)) for q in (0, 5, 50, 95, 100)]
res["corr_simd_recs_end"] = float(np.corrcoef(per_simd_recs, per_simd_end)[0, 1])
print(json.dumps(res))
