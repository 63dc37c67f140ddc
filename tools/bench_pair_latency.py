"""Latency of the reference's pairwise entry points through the legacy ABI (steady state)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
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
