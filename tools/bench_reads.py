"""Short-read workload: N records of 150 bases in one device buffer (the shape of a FASTQ batch)."""
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
rl = int(sys.argv[1]) if len(sys.argv) > 1 else 150
nrec = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
n = rl * nrec
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, C.c_void_p(0))
torch.cuda.synchronize()
off = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(rl)
for params, name, force in [((0, 31, False, 42, 18446744073709552, True), "scaled=1000 abund", True),
                            ((0, 31, False, 42, 18446744073709552, True), "scaled=1000 abund, force=false", False),
                            ((1000, 21, False, 42, 0, False), "num=1000 k=21", True),
                            ((0, 27, True, 42, 18446744073709552, True), "protein ksize=27 scaled=1000 abund", True)]:
    L.smh_profile_reset(); L.smh_profile_enable(1)
    ts = []
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mh = pkg.KmerMinHash(*params)
        mh.add_sequences_dev(buf.data_ptr(), n, off, force)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"protein_fused" if params[2] else b"dna_rolling", C.byref(ms), C.byref(cnt))
    kmers = nrec * (rl - params[1] + 1) * (2 if params[2] else 1)     # protein: two windows per start position
    print("%d reads x %d bp, %s: total %.1f ms (%.1f G k-mers/s), kernel time %.1f ms over %d launches, sketch %d" % (
        nrec, rl, name, min(ts) * 1e3, kmers / min(ts) / 1e9, ms.value / 3, cnt.value / 3, len(mh)))
