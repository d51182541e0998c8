"""Debug aid: where does the end-to-end gradient leg exceed its allowance?  (GPU box)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests import test_gpu_render as TR, helpers as H
import brush_amd.render as R

dev = torch.device("cuda:0")
n, w, h, deg, mult = [float(x) if "." in x else int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (20000, 256, 192, 0, 0.01)
cloud = H.synthetic_cloud(n, deg, seed=4, mean_mult=mult)
gpu, orc = TR._run_pair(dev, cloud, w, h, deg, max_intersects=4_000_000)
f64 = orc["grads_f64_e2e"]
for name in TR.GRAD_NAMES:
    t = f64[name]
    a = gpu[name].detach().cpu().numpy().astype(np.float64).reshape(t.shape)
    ref_err = np.abs(orc["grads"][name].astype(np.float64).reshape(t.shape) - t)
    ratio, err, parts = TR._grad_ratio(a, f64, TR._rowmax(ref_err, t.shape), name, TR.CANDIDATES[TR.ACTIVE])
    i = np.unravel_index(np.argmax(ratio), ratio.shape)
    print(name, "worst", ratio[i], "at", i, "err", err[i], "t", t[i], "mag", f64["mag_" + name[2:]][i], "flip", f64["flip_" + name[2:]][i])
    if name == "v_means":
        g = i[0]
        oa = orc["aux"]
        V = int(oa["num_visible"][0])
        c = int(np.nonzero(oa["global_from_compact_gid"][:V] == g)[0][0])
        p = oa["projected_splats"][c]
        print("splat", g, "compact", c, "proj", p)
        fi_g = TR._np_u32(gpu["aux"].final_index)
        fi_o = oa["final_index"]
        T_g = 1.0 - gpu["out"][..., 3].astype(np.float64)
        T_o = 1.0 - orc["out"][..., 3].astype(np.float64)
        wgt = 2.0 * np.abs(T_g - T_o) / np.maximum(np.minimum(T_g, T_o), 1e-5) + (fi_g != fi_o)
        x0, x1, y0, y1 = 0, w, 0, h
        sub = wgt[y0:y1, x0:x1]
        print("weights near splat: max", sub.max(), "count>1e-5", int((sub > 1e-5).sum()), "fin differ", int((fi_g != fi_o)[y0:y1, x0:x1].sum()),
              "risk", int(oa["flip_risk"][y0:y1, x0:x1].sum()))
        print("weight quantiles", np.quantile(sub, [0.5, 0.9, 0.99, 0.999, 1.0]), "sum", sub.sum())
        order = np.argsort(sub.reshape(-1))[::-1][:12]
        ys, xs = np.unravel_index(order, sub.shape)
        for yy, xx in list(zip(ys, xs))[:12]:
            Y, X = yy + y0, xx + x0
            print("  px", X, Y, "w", wgt[Y, X], "T_g", T_g[Y, X], "T_o", T_o[Y, X], "fin", fi_g[Y, X], fi_o[Y, X], "risk", oa["flip_risk"][Y, X],
                  "out_g", gpu["out"][Y, X], "out_o", orc["out"][Y, X])
        # shared-state arbiter for the same element
        s = orc["grads_f64"]
        print("shared-state f64:", s[name][i], "e2e f64:", t[i], "gpu:", a[i], "oracle f32 e2e:", orc["grads"][name][i], "oracle f32 shared", orc["grads_f32"][name][i])
