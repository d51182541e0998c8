import torch, time, sys
sys.path.insert(0, ".")
import math
from brush_amd.synthetic import synthetic_cloud
from brush_amd import dist as BD, render as R
import brush_amd
dev = torch.device("cuda:0")
n, w, h, deg = 1<<20, 1920, 1080, 3
C = 16
cloud = synthetic_cloud(n, deg, seed=4)
p = {k: torch.from_numpy(v).to(dev) for k, v in cloud.items()}
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)  # render_bench.rs:163-174: (0,0,-8), fov 90 deg on x
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, brush_amd._lib.lib().brush_default_max_intersects(n, w, h))
v_out = torch.randn((h, w, 4), device=dev) / (h*w)
g, block = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out)
V = aux.read_num_visible()
rows = -(-V // 256) * 256
print("V", V)
def t(f, reps=20):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
print("pack hip us", t(lambda: BD.pack_view_records(block, aux, n, C, rows)))
rec = BD.pack_view_records(block, aux, n, C, rows)
camw = aux.uniforms_buffer[12:15].contiguous().view(torch.float32)[None]
for W in (2, 4, 8):
    recs = rec[None].repeat(W, 1, 1).contiguous()
    cnt = torch.full((W,), V, dtype=torch.int32, device=dev)
    b2 = block.clone()
    print(W, "expand hip us", t(lambda: BD.expand_view_records(recs, cnt, camw.repeat(W, 1), p["means"], b2, n, C, own_view=0)))
