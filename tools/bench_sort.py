"""smh_sort_u64 on a few key distributions, mode 0 (plain passes) / 1 (hashed keys); the time includes
the copies to and from the device (the same for every mode).  python tools/bench_sort.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
rng = np.random.default_rng(1)
cases = {
    "25 M distinct scaled hashes": rng.integers(0, 18446744073709552, size=25_000_000, dtype=np.uint64),
    "5 M distinct scaled hashes": rng.integers(0, 18446744073709552, size=5_000_000, dtype=np.uint64),
    "5 M keys = 166 k hashes x 30 copies": np.repeat(rng.integers(0, 18446744073709552, size=166_667, dtype=np.uint64), 30),
    "5 M keys, one of them 1 M times": np.concatenate([rng.integers(0, 2**55, size=4_000_000, dtype=np.uint64), np.full(1_000_000, 12345678901234567, dtype=np.uint64)]),
    "25 M keys = 830 k hashes x 30 copies": np.repeat(rng.integers(0, 18446744073709552, size=833_334, dtype=np.uint64), 30)[:25_000_000],
    "20 M pooled: 200 k hashes x 70 + 6 M singles": np.concatenate([np.repeat(rng.integers(0, 2**64, size=200_000, dtype=np.uint64), 70),
                                                                   rng.integers(0, 2**64, size=6_000_000, dtype=np.uint64)]),
    "10 M keys, one of them 1 M times": np.concatenate([rng.integers(0, 2**55, size=9_000_000, dtype=np.uint64), np.full(1_000_000, 12345678901234567, dtype=np.uint64)]),
}
for name, keys in cases.items():
    rng.shuffle(keys)
    want = np.sort(keys)
    row = []
    for mode in (0, 1):
        best = 1e9
        for _ in range(3):
            k = keys.copy()
            p = np.arange(k.size, dtype=np.uint32)
            t0 = time.perf_counter()
            assert L.smh_sort_u64(k.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p), k.size, mode) == 0
            best = min(best, time.perf_counter() - t0)
        assert np.array_equal(k, want)
        row.append(best * 1e3)
    print("%-48s mode 0 %7.1f ms   mode 1 %7.1f ms" % (name, row[0], row[1]), flush=True)
