"""k-mer size sweep on 2 GB (k=31/21/51 are compile-time instantiations, the rest run-time k)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
n = 2_000_000_000
buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, None) == 0
torch.cuda.synchronize()
off = np.array([0, n], dtype=np.uint64)
MAXH = 18446744073709552
for k in [int(x) for x in (sys.argv[1:] or "15 21 25 27 31 32 33 40 51 63 64 100 128".split())]:
    for it in range(3):
        mh = pkg.KmerMinHash(0, k, False, 42, MAXH, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mh.add_sequences_dev(buf.data_ptr(), n, off, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("k=%-4d %.2f ms  %.1f G k-mers/s" % (k, dt * 1e3, n / dt / 1e9), flush=True)
