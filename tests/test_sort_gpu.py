"""The library's own stable radix sort (libwave_amd/csrc/wm_sort.hpp: k_rs_hist / k_rs_scan / k_rs_scatter -- what every
cloud's Morton order, the NDT voxel order and the voxel filter are built on above 256k points) against numpy's stable
argsort: tile boundaries, every pass count (odd and even: the ping-pong ends in different buffers), few and many equal
keys, 32- and 64-bit keys, bits that are not a multiple of the digit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _expect(keys, bits):
    mask = (1 << bits) - 1 if bits < 64 else (1 << 64) - 1
    low = (keys.astype(np.uint64) & np.uint64(mask))
    return np.argsort(low, kind="stable").astype(np.uint32)


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 4095, 4096, 4097, 8192, 65537, 300_001, 1_000_000])
@pytest.mark.parametrize("bits", [1, 7, 8, 9, 16, 22, 24, 32])
def test_sort_pairs_u32_is_numpys_stable_argsort(ctx, n, bits):
    rng = np.random.default_rng(1000 * bits + n % 997)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    got = ctx.sort_pairs(keys, bits)
    assert np.array_equal(got, _expect(keys, bits))


@pytest.mark.parametrize("bits", [5, 24, 33, 40, 63, 64])
def test_sort_pairs_u64(ctx, bits):
    rng = np.random.default_rng(bits)
    n = 123_457
    keys = rng.integers(0, np.iinfo(np.uint64).max, n, dtype=np.uint64, endpoint=True)
    got = ctx.sort_pairs(keys, bits)
    assert np.array_equal(got, _expect(keys, bits))


def test_sort_pairs_many_equal_keys_and_sorted_inputs(ctx):
    n = 500_000
    rng = np.random.default_rng(5)
    for keys in (np.zeros(n, np.uint32), np.arange(n, dtype=np.uint32), np.arange(n, dtype=np.uint32)[::-1].copy(),
                 rng.integers(0, 3, n).astype(np.uint32), (np.arange(n, dtype=np.uint32) // 5000).astype(np.uint32)):
        got = ctx.sort_pairs(keys, 22)
        assert np.array_equal(got, _expect(keys, 22))
