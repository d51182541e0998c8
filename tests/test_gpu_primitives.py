"""GPU parity of the sort / scan primitives through the C ABI (pytest -m gpu).

Vectors of crates/brush-sort/src/lib.rs:164-265 and crates/brush-prefix-sum/src/lib.rs:110-175,
plus the oracle (oracle/brush_oracle.c) on seeded inputs and size-independent properties.
"""
import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import brush_amd.render as R

    R.DEBUG_POISON = True
    return torch.device("cuda:0")


def _sort_gpu(keys, vals, n_sort, bits, dev):
    import torch

    import brush_amd

    k = torch.as_tensor(keys.view(np.int32), device=dev)
    v = torch.as_tensor(vals.view(np.int32), device=dev)
    n = torch.tensor([n_sort], dtype=torch.int32, device=dev)
    ko, vo = brush_amd.radix_argsort(k, v, n, bits)
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32)


def test_sorting_reference_vectors(dev):
    for i in range(0, 128, 7):
        keys = np.array([5 + i * 4, i, 6, 123, 74657, 123, 999, 2 ** 24 + 123, 6, 7, 8, 0, i * 2, 16 + i, 128 * i],
                        np.uint32)
        vals = keys * 2 + 5
        ko, vo = _sort_gpu(keys, vals, len(keys), 32, dev)
        idx = np.argsort(keys.astype(np.int64), kind="stable")
        assert np.array_equal(ko, keys[idx]) and np.array_equal(vo, vals[idx])


def test_sorting_big_clustered(dev):
    rng = np.random.default_rng(0)
    chunks = []
    for i in range(10000):
        start = rng.integers(i, i + 150)
        end = rng.integers(start, start + 250)
        js = np.arange(start, end)
        chunks.append(js[rng.random(len(js)) < 0.5])
    keys = np.concatenate(chunks).astype(np.uint32)
    vals = keys * 2 + 5
    ko, vo = _sort_gpu(keys, vals, len(keys), 32, dev)
    idx = np.argsort(keys.astype(np.int64), kind="stable")
    assert np.array_equal(ko, keys[idx]) and np.array_equal(vo, vals[idx])


@pytest.mark.parametrize("n,bits", [(1, 32), (63, 32), (64, 8), (4096, 13), (4097, 13), (100000, 10), (100000, 4),
                                    (100000, 1), (1 << 20, 32), (3000000, 16), (5000, 0), (65536, 32), (65537, 32),
                                    (12_000_000, 16)])  # the last one: > 512 tiles of 16384 keys, the 3-launch shape
def test_sort_matches_oracle(dev, n, bits):
    """Stability and the 'low 4*ceil(bits/4) bits only' rule vs the oracle, ragged sizes."""
    rng = np.random.default_rng(n + bits)
    keys = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    if bits >= 13:  # many duplicates to exercise stability
        keys[: n // 2] &= np.uint32(0x1FFF)
    vals = np.arange(n, dtype=np.uint32)
    ko, vo = _sort_gpu(keys, vals, n, bits, dev)
    rk, rv = O.radix_argsort(keys, vals, bits=bits)
    assert np.array_equal(ko, rk) and np.array_equal(vo, rv)


def test_sort_partial_count(dev):
    """n_sort < len: only the first n_sort pairs are sorted (sort_scatter.wgsl:118-121)."""
    rng = np.random.default_rng(5)
    keys = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(20000, dtype=np.uint32)
    for n_sort in (0, 1, 4095, 4096, 12345):
        ko, vo = _sort_gpu(keys, vals, n_sort, 32, dev)
        idx = np.argsort(keys[:n_sort].astype(np.int64), kind="stable")
        assert np.array_equal(ko[:n_sort], keys[:n_sort][idx]) and np.array_equal(vo[:n_sort], vals[:n_sort][idx])


def test_sort_large_properties(dev):
    """8.39 M pairs (the reference's intersection cap): sortedness + permutation + stability."""
    import torch

    import brush_amd

    n = 128 * 65535
    g = torch.Generator(device=dev).manual_seed(1)
    keys = torch.randint(0, 8160, (n,), dtype=torch.int32, device=dev, generator=g)
    vals = torch.arange(n, dtype=torch.int32, device=dev)
    ko, vo = brush_amd.radix_argsort(keys, vals, torch.tensor([n], dtype=torch.int32, device=dev), 13)
    assert bool((ko[1:] >= ko[:-1]).all())
    assert bool((keys[vo.long()] == ko).all())
    same = ko[1:] == ko[:-1]
    assert bool((vo[1:][same] > vo[:-1][same]).all())  # stable: ties keep input order
    assert int(torch.bincount(vo.long(), minlength=n).max()) == 1


def test_sort_rejects_bad_bits(dev):
    import torch

    import brush_amd

    k = torch.zeros(4, dtype=torch.int32, device=dev)
    with pytest.raises(AssertionError):
        brush_amd.radix_argsort(k, k.clone(), torch.tensor([4], dtype=torch.int32, device=dev), 33)


def _scan_gpu(x, dev):
    import torch

    import brush_amd

    return brush_amd.prefix_sum(torch.as_tensor(x.view(np.int32), device=dev)).cpu().numpy().view(np.uint32)


def test_prefix_sum_reference_vectors(dev):
    assert np.array_equal(_scan_gpu(np.array([1, 1, 1, 1], np.uint32), dev), [1, 2, 3, 4])
    data = 90 + np.arange(1024, dtype=np.uint32)
    assert np.array_equal(_scan_gpu(data, dev), np.cumsum(data, dtype=np.uint64).astype(np.uint32))
    n = 512 * 16 + 123
    data = np.stack([2 + np.arange(n), np.zeros(n), np.full(n, 32), np.full(n, 512), np.full(n, 30965)],
                    axis=1).reshape(-1).astype(np.uint32)
    assert np.array_equal(_scan_gpu(data, dev), np.cumsum(data, dtype=np.uint64).astype(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 3, 5, 1023, 1024, 1025, 4099, 1 << 20, (1 << 20) + 7, 20_971_520])
def test_prefix_sum_sizes(dev, n):
    rng = np.random.default_rng(n)
    data = rng.integers(0, 50, n, dtype=np.uint32)
    assert np.array_equal(_scan_gpu(data, dev), np.cumsum(data, dtype=np.uint64).astype(np.uint32))


def test_prefix_sum_wraps(dev):
    data = np.full(10, 0x40000000, np.uint32)
    assert np.array_equal(_scan_gpu(data, dev), O.inclusive_scan(data))
