"""Randomised sweep of the sort / scan primitives on the GPU box (development aid, not collected by pytest):
random sizes (across the tile-size and launch-shape switches of radix_sort.hip), bit counts, key distributions and
partial counts against numpy.  usage: python tests/fuzz_primitives.py [cases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import brush_amd  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    for i in range(cases):
        n = int(rng.choice([1, 2, 63, 1023, 1024, 1025, 4097, 131071, 131073, 262145, 600000, 1100000, 2200000, 4300000,
                            8388480, 8388700, 9000000, 12582913]))
        n = max(1, n + int(rng.integers(-3, 4)))
        bits = int(rng.choice([0, 1, 4, 7, 8, 9, 13, 15, 16, 17, 24, 31, 32]))
        kind = int(rng.integers(0, 4))
        if kind == 0:
            keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        elif kind == 1:
            keys = rng.integers(0, 16, n, dtype=np.uint64).astype(np.uint32)  # few distinct keys: long equal runs
        elif kind == 2:
            keys = np.sort(rng.integers(0, 2 ** 20, n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
        else:
            keys = (np.arange(n, dtype=np.uint64) * 2654435761 % (2 ** 32)).astype(np.uint32)
        vals = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
        n_sort = n if rng.random() < 0.6 else int(rng.integers(0, n + 1))
        k = torch.as_tensor(keys.view(np.int32), device=dev)
        v = torch.as_tensor(vals.view(np.int32), device=dev)
        cnt = torch.tensor([n_sort], dtype=torch.int32, device=dev)
        ko, vo = brush_amd.radix_argsort(k, v, cnt, bits)
        ko, vo = ko.cpu().numpy().view(np.uint32)[:n_sort], vo.cpu().numpy().view(np.uint32)[:n_sort]
        total_bits = 4 * ((bits + 3) // 4)  # brush-sort/src/lib.rs:58
        mask = np.uint64((1 << total_bits) - 1)
        idx = np.argsort((keys[:n_sort].astype(np.uint64) & mask), kind="stable")
        assert np.array_equal(ko, keys[:n_sort][idx]) and np.array_equal(vo, vals[:n_sort][idx]), (i, n, bits, kind, n_sort)
        # scan
        m = int(rng.choice([1, 255, 256, 257, 65537, 1 << 20, (1 << 20) + 3, 3_000_001]))
        x = rng.integers(0, 50, m, dtype=np.uint64).astype(np.uint32)
        got = brush_amd.prefix_sum(torch.as_tensor(x.view(np.int32), device=dev)).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, np.cumsum(x.astype(np.uint64)).astype(np.uint32)), (i, m)
        print(f"case {i}: sort n={n} n_sort={n_sort} bits={bits} kind={kind}; scan m={m} ok", flush=True)
    print("all", cases, "cases ok")


if __name__ == "__main__":
    main()
