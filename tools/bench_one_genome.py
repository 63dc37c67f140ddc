"""One genome per call: a 5 Mbp record resident in HBM sketched with scaled=1000, k=31, fifty times (the per-call latency of
the shape `sourmash sketch` has: launch, fold, synchronisations).  python tools/bench_one_genome.py [bases]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 9, 0, None) == 0
torch.cuda.synchronize()
off = np.array([0, n], dtype=np.uint64)
for params, name in (((0, 31, False, 42, 18446744073709552, False), "scaled=1000"), ((0, 31, False, 42, 18446744073709552, True), "scaled=1000 abund"),
                     ((1000, 31, False, 42, 0, False), "num=1000"), ((1000, 31, False, 42, 0, True), "num=1000 abund"),
                     ((0, 30, True, 42, 18446744073709552, False), "protein ksize=30 scaled=1000")):
    ts = []
    for it in range(50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mh = pkg.KmerMinHash(*params)
        mh.add_sequences_dev(buf.data_ptr(), n, off, True)
        size = len(mh)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print("%d bases, %s: %.3f ms per sketch (median of 50), |sketch| %d" % (n, name, sorted(ts)[25] * 1e3, size), flush=True)
