"""Latency of the reference's pairwise entry points through the legacy ABI (steady state)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
rng = np.random.RandomState(1)
for num in (500, 2000):
    a, b = pkg.KmerMinHash(num, 31, False, 42, 0, False), pkg.KmerMinHash(num, 31, False, 42, 0, False)
    pool = np.unique(rng.randint(0, 1 << 62, size=3 * num, dtype=np.int64).astype(np.uint64))
    a.add_many(pool[::2]); b.add_many(pool[::3])
    for name, fn in (("compare", lambda: a.compare(b)), ("count_common", lambda: a.count_common(b)),
                     ("intersection_size", lambda: a.intersection_size(b))):
        for _ in range(20): fn()
        t0 = time.perf_counter()
        for _ in range(500): r = fn()
        dt = (time.perf_counter() - t0) / 500
        print("num=%d %-18s %.1f us per call -> %s" % (num, name, dt * 1e6, r))

# two large scaled sketches built on the device (2 GB of DNA each, half of it shared)
import ctypes as C
L = pkg.lib()
n = 2_000_000_000
buf = torch.empty(n + n // 2 + 64, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n + n // 2, 2, 0, None) == 0
MAXH = 18446744073709552
off = np.array([0, n], dtype=np.uint64)
a = pkg.KmerMinHash(0, 31, False, 42, MAXH, False); a.add_sequences_dev(buf.data_ptr(), n, off, True)
b = pkg.KmerMinHash(0, 31, False, 42, MAXH, False); b.add_sequences_dev(buf.data_ptr() + n // 2, n, off, True)
t0 = time.perf_counter(); j = a.compare(b); t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(20): j = a.compare(b)
dt = (time.perf_counter() - t0) / 20
print("scaled sketches of %d and %d hashes: first compare (brings both to the host) %.1f ms, then %.3f ms per call -> %.4f, common %d"
      % (len(a), len(b), t_first * 1e3, dt * 1e3, j, a.count_common(b)))
