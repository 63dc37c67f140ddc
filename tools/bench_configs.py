"""Secondary timings of the other BASELINE configs on one GPU (not the bench line): C1 latency,
num-mode throughput, k=21 / k=51 DNA, protein arm with abundance.  Prints one line per case."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
MAXH = 18446744073709552


def dev_dna(n, seed=2):
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, seed, 0, C.c_void_p(0))
    torch.cuda.synchronize()
    return buf


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


big = dev_dna(2_000_000_000)
cases = [
    ("C1  1 MB k=31 num=500 (host bytes in, legacy ABI)", None),
    ("DNA 2 GB k=31 num=500", (500, 31, False, 42, 0, False)),
    ("DNA 2 GB k=31 num=2000 abund", (2000, 31, False, 42, 0, True)),
    ("DNA 2 GB k=21 scaled=1000", (0, 21, False, 42, MAXH, False)),
    ("DNA 2 GB k=25 scaled=1000 (run-time k)", (0, 25, False, 42, MAXH, False)),
    ("DNA 0.5 GB k=51 scaled=1000 (byte-wise kernel)", (0, 51, False, 42, MAXH, False)),
    ("PROT 0.5 GB ksize=27 scaled=1000 abund", (0, 27, True, 42, MAXH, True)),
]
host1m = bytes(big[:1_000_000].cpu().numpy())
for name, params in cases:
    if params is None:
        def run():
            mh = pkg.KmerMinHash(500, 31)
            mh.add_sequence(host1m, True)
            len(mh)          # observing the sketch runs the queued batch
            return mh
        dt = timeit(run, 10)
        mh = run()
        t0 = time.perf_counter(); j = mh.compare(mh); dtc = time.perf_counter() - t0
        print("%-52s %.3f ms per sketch (%.2f G k-mers/s), self-compare %.3f ms -> %.1f" % (name, dt * 1e3, 1e6 / dt / 1e9, dtc * 1e3, j))
        continue
    n = 500_000_000 if ("0.5 GB" in name) else 2_000_000_000
    off = np.array([0, n], dtype=np.uint64)

    def run():
        mh = pkg.KmerMinHash(*params)
        mh.add_sequences_dev(big.data_ptr(), n, off, True)
        return mh
    dt = timeit(run, 2)
    mh = run()
    units = n * (2 if params[2] else 1)
    print("%-52s %.2f ms  %.1f G %s/s  (sketch size %d)" % (name, dt * 1e3, units / dt / 1e9, "windows" if params[2] else "k-mers", len(mh)))
