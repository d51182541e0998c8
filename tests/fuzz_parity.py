"""Randomised parity sweep on the GPU box (development aid, not part of the suite): small random clouds, image sizes,
SH degrees and spreads through the same GPU-vs-checker assertions as tests/test_gpu_render.py
(bit-exact integer stages, pixels, arbiter-based gradient tolerance).  usage: python tests/fuzz_parity.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from tests import helpers as H  # noqa: E402
from tests import test_gpu_render as T  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    for i in range(cases):
        n = int(rng.choice([1, 2, 63, 64, 65, 200, 1000, 5000, 20000, 70000]))
        w, h = int(rng.integers(1, 500)), int(rng.integers(1, 400))
        deg = int(rng.integers(0, 5))
        mult = float(rng.choice([0.0005, 0.002, 0.01, 0.05, 0.3, 1.0]))
        seed = int(rng.integers(0, 1000))
        cap = int(rng.choice([3_000_000, 3_000_000, 20_000, 700]))  # small capacities: truncated lists, overflow flag
        tag = f"case {i}: n={n} {w}x{h} deg={deg} mult={mult} seed={seed} cap={cap}"
        cloud = H.synthetic_cloud(n, deg, seed=seed, mean_mult=mult)
        gpu, orc = T._run_pair(dev, cloud, w, h, deg, max_intersects=cap)
        V, I = T._assert_forward_parity(gpu, orc, w, h)
        T._assert_grad_parity(gpu, orc, tag)
        print(tag, "V", V, "I", I, "ok", flush=True)
    print("all", cases, "cases ok")


if __name__ == "__main__":
    main()
