"""Unit pins for the oracle's sort / scan (the reference's own vectors) and detmath."""
import numpy as np

from oracle import oracle as O


def _argsort_ref(keys):
    return np.argsort(np.asarray(keys, np.int64), kind="stable")


def test_sorting_reference_vectors():
    """brush-sort/src/lib.rs:164-216: 15 keys x 128 variants, 32-bit, vs CPU stable argsort."""
    for i in range(128):
        keys = np.array([5 + i * 4, i, 6, 123, 74657, 123, 999, 2 ** 24 + 123, 6, 7, 8, 0, i * 2,
                         16 + i, 128 * i], np.uint32)
        vals = keys * 2 + 5
        ko, vo = O.radix_argsort(keys, vals, bits=32)
        idx = _argsort_ref(keys)
        assert np.array_equal(ko, keys[idx]) and np.array_equal(vo, vals[idx])


def test_sorting_big_clustered():
    """brush-sort/src/lib.rs:218-265 (seeded here; the reference uses thread_rng)."""
    rng = np.random.default_rng(0)
    chunks = []
    for i in range(10000):
        start = rng.integers(i, i + 150)
        end = rng.integers(start, start + 250)
        js = np.arange(start, end)
        chunks.append(js[rng.random(len(js)) < 0.5])
    keys = np.concatenate(chunks).astype(np.uint32)
    vals = keys * 2 + 5
    ko, vo = O.radix_argsort(keys, vals, bits=32)
    idx = _argsort_ref(keys)
    assert np.array_equal(ko, keys[idx]) and np.array_equal(vo, vals[idx])


def test_sort_low_bits_only_and_stable():
    """Only the low 4*ceil(bits/4) bits take part (brush-sort/src/lib.rs:58), order stable."""
    rng = np.random.default_rng(1)
    keys = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(5000, dtype=np.uint32)
    ko, vo = O.radix_argsort(keys, vals, bits=13)
    idx = np.argsort((keys & 0xFFFF).astype(np.int64), kind="stable")
    assert np.array_equal(ko, keys[idx]) and np.array_equal(vo, vals[idx])
    ko, vo = O.radix_argsort(keys, vals, n_sort=100, bits=32)
    idx = _argsort_ref(keys[:100])
    assert np.array_equal(ko[:100], keys[:100][idx])


def test_prefix_sum_reference_vectors():
    """brush-prefix-sum/src/lib.rs:110-175."""
    assert np.array_equal(O.inclusive_scan([1, 1, 1, 1]), [1, 2, 3, 4])
    data = 90 + np.arange(1024, dtype=np.uint32)
    assert np.array_equal(O.inclusive_scan(data), np.cumsum(data, dtype=np.uint64).astype(np.uint32))
    n = 512 * 16 + 123
    data = np.stack([2 + np.arange(n), np.zeros(n), np.full(n, 32), np.full(n, 512),
                     np.full(n, 30965)], axis=1).reshape(-1).astype(np.uint32)
    assert len(data) == 41575
    assert np.array_equal(O.inclusive_scan(data), np.cumsum(data, dtype=np.uint64).astype(np.uint32))


def _ulp_diff(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(a - b)


def test_detmath_close_to_libm():
    xs = np.concatenate([np.linspace(-20, 20, 20001), np.linspace(-87, 88, 5001)]).astype(np.float32)
    got = np.array([O.det_expf(float(x)) for x in xs], np.float32)
    want = np.exp(xs.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(got, want).max() <= 2
    xs = np.concatenate([np.linspace(1e-6, 4, 20001), np.geomspace(1e-30, 1e30, 5001)]).astype(np.float32)
    got = np.array([O.det_logf(float(x)) for x in xs], np.float32)
    want = np.log(xs.astype(np.float64)).astype(np.float32)
    big = np.abs(want) > 1e-3
    assert _ulp_diff(got[big], want[big]).max() <= 2
    assert np.abs(got[~big].astype(np.float64) - np.log(xs[~big].astype(np.float64))).max() < 1e-9
    assert O.det_expf(0.0) == 1.0 and O.det_logf(1.0) == 0.0
    assert O.det_expf(-200.0) == 0.0 and np.isinf(O.det_expf(100.0))
