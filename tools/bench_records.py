"""Does the record table cost anything in the hot kernel?  Same 10 GB, k=31 scaled=1000, cut into
1 / 10 000 / 1 000 000 records.  python tools/bench_records.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000_000
buf = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 2, 0, None) == 0
torch.cuda.synchronize()
MAXH = 18446744073709552
for nrec in (1, 10_000, 1_000_000):
    off = (np.arange(nrec + 1, dtype=np.uint64) * np.uint64(total // nrec))
    off[-1] = total
    for it in range(3):
        mh = pkg.KmerMinHash(0, 31, False, 42, MAXH, False)
        L.smh_profile_reset(); L.smh_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mh.add_sequences_dev(buf.data_ptr(), total, off, True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        L.smh_profile_enable(0)
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"dna_rolling", C.byref(ms), C.byref(cnt))
    print("%8d records: call %.2f ms, kernel %.2f ms (%.1f G k-mers/s), sketch %d" % (nrec, dt * 1e3, ms.value, total / ms.value / 1e6, len(mh)), flush=True)
