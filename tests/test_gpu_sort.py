"""The fold's sort on its own (smh_sort_u64) against numpy's stable sort: uniform hashes, scaled hashes, repeated keys
(a k-mer a million times, every key thirty times, the pool of one family), keys that share their high bits, constant
high bytes, short keys (passes skipped), the sizes around the one-workgroup sort."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sort(pkg, keys, payload):
    k = np.ascontiguousarray(keys, dtype=np.uint64).copy()
    p = None if payload is None else np.ascontiguousarray(payload, dtype=np.uint32).copy()
    rc = pkg.lib().smh_sort_u64(k.ctypes.data_as(C.c_void_p), None if p is None else p.ctypes.data_as(C.c_void_p), k.size)
    assert rc == 0
    return k, p


def _check(pkg, keys):
    keys = np.asarray(keys, dtype=np.uint64)
    order = np.argsort(keys, kind="stable")
    want = keys[order]
    k, _ = _sort(pkg, keys, None)
    assert np.array_equal(k, want), "keys only"
    k, p = _sort(pkg, keys, np.arange(keys.size, dtype=np.uint32))
    assert np.array_equal(k, want), "keys with payload"
    assert np.array_equal(p, order.astype(np.uint32)), "payload order (stability)"


@pytest.mark.parametrize("n", [2, 63, 64, 65, 511, 512, 513, 4096, 4097, 8191, 8192, 8193, 65535, 65536, 300_000, 3_000_000])
def test_uniform_keys(pkg, n):
    rng = np.random.default_rng(n)
    _check(pkg, rng.integers(0, 2**64, size=n, dtype=np.uint64))


@pytest.mark.parametrize("n", [70_000, 1_000_000])
def test_keys_under_a_scaled_threshold(pkg, n):
    rng = np.random.default_rng(n + 1)
    _check(pkg, rng.integers(0, 18446744073709552, size=n, dtype=np.uint64))   # max_hash of scaled=1000: 55 bits


def test_repeated_keys_make_big_equal_buckets(pkg):
    rng = np.random.default_rng(7)
    distinct = rng.integers(0, 2**64, size=300, dtype=np.uint64)
    _check(pkg, distinct[rng.integers(0, distinct.size, size=400_000)])          # ~1300 copies of each key
    _check(pkg, np.full(10_000, 0x123456789ABCDEF0, dtype=np.uint64))            # one key: nothing to sort


def test_keys_that_share_their_high_bits(pkg):
    rng = np.random.default_rng(8)
    low = rng.integers(0, 2**48, size=100_000, dtype=np.uint64)
    _check(pkg, (np.uint64(0x5A5A) << np.uint64(48)) | low)                      # ONE bucket of 100 000 different keys
    # a uniform background, one bucket of 5 000 different keys, one run of 2 000 equal keys
    bg = rng.integers(0, 2**64, size=200_000, dtype=np.uint64)
    heavy = (np.uint64(0x0123) << np.uint64(48)) | rng.integers(0, 2**48, size=5_000, dtype=np.uint64)
    run = np.full(2_000, 0xFEDC_0000_0000_0001, dtype=np.uint64)
    mix = np.concatenate([bg, heavy, run])
    rng.shuffle(mix)
    _check(pkg, mix)


def test_constant_high_byte_and_short_keys(pkg):
    rng = np.random.default_rng(9)
    _check(pkg, (np.uint64(0xAB) << np.uint64(56)) | rng.integers(0, 2**40, size=150_000, dtype=np.uint64))
    _check(pkg, rng.integers(0, 2**20, size=150_000, dtype=np.uint64))           # 20-bit keys, many repeats
    _check(pkg, rng.integers(0, 2**33, size=150_000, dtype=np.uint64))


def test_one_key_repeated_a_million_times_among_distinct_ones(pkg):
    rng = np.random.default_rng(10)
    keys = np.concatenate([rng.integers(0, 2**55, size=500_000, dtype=np.uint64), np.full(1_000_000, 0x0012_3456_789A_BCDE, dtype=np.uint64),
                           np.full(70, 0x0000_0000_0000_0007, dtype=np.uint64)])
    rng.shuffle(keys)
    _check(pkg, keys)


def test_every_key_thirty_times(pkg):
    """reads at 30-fold coverage: every retained hash is a long run of equal keys"""
    rng = np.random.default_rng(11)
    distinct = rng.integers(0, 18446744073709552, size=100_000, dtype=np.uint64)
    keys = np.repeat(distinct, 30)
    rng.shuffle(keys)
    _check(pkg, keys)


def test_pool_of_one_family(pkg):
    """every pool hash in most signatures of the block: 4 000 keys x 800-2 300 copies beside a million single ones"""
    rng = np.random.default_rng(12)
    for copies in (800, 2300):
        heavy = np.repeat(rng.integers(0, 2**64, size=1500, dtype=np.uint64), copies)
        keys = np.concatenate([heavy, rng.integers(0, 2**64, size=1_000_000, dtype=np.uint64)])
        rng.shuffle(keys)
        _check(pkg, keys)


def test_small_arrays_with_repeats(pkg):
    """the one-workgroup sort of up to 8192 keys: ties keep their order, constant digits are skipped"""
    rng = np.random.default_rng(13)
    for n in (5, 100, 2000, 5000, 8192):
        _check(pkg, rng.integers(0, 50, size=n, dtype=np.uint64))
        _check(pkg, np.full(n, 7, dtype=np.uint64))
        _check(pkg, np.full(n, 2**64 - 1, dtype=np.uint64))
        _check(pkg, rng.integers(2**63, 2**64, size=n, dtype=np.uint64))
