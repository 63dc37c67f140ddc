"""Where the protein kernel's time goes: the C5 share with (a) scaled=1000, (b) a threshold nothing passes
(no emit path), (c) one record instead of 12 500 (no record boundaries), (d) DNA kernel on the same bytes for
comparison.  python tools/prot_variants.py [n_records]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
rlen = 1_000_000
total = nrec * rlen
buf = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 5, 0, None) == 0
torch.cuda.synchronize()
off = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(rlen)
one = np.array([0, total], dtype=np.uint64)

def run(label, prot, ksize, maxh, offsets, kern):
    def go():
        mh = pkg.KmerMinHash(0, ksize, prot, 42, maxh, prot)
        mh.add_sequences_dev(buf.data_ptr(), total, offsets, True)
        return mh
    go(); torch.cuda.synchronize()
    L.smh_profile_reset(); L.smh_profile_enable(1)
    t0 = time.perf_counter(); mh = go(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    L.smh_profile_enable(0)
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(kern, C.byref(ms), C.byref(cnt))
    print("%-52s step %7.2f ms  kernel %7.2f ms (%d launches)  |sketch| %d" % (label, dt * 1e3, ms.value, cnt.value, len(mh)), flush=True)

MAXH = 18446744073709552
run("protein k=27 scaled=1000, 1 MB records", True, 27, MAXH, off, b"protein_fused")
run("protein k=27 max_hash=1 (nothing passes)", True, 27, 1, off, b"protein_fused")
run("protein k=27 scaled=1000, ONE record", True, 27, MAXH, one, b"protein_fused")
run("protein k=21 scaled=1000, 1 MB records", True, 21, MAXH, off, b"protein_fused")
run("protein k=30 scaled=1000, 1 MB records", True, 30, MAXH, off, b"protein_fused")
run("DNA k=31 scaled=1000, 1 MB records", False, 31, MAXH, off, b"dna_rolling")
run("DNA k=31 max_hash=1", False, 31, 1, off, b"dna_rolling")
run("DNA k=31 scaled=1000, ONE record", False, 31, MAXH, one, b"dna_rolling")
run("DNA k=21 scaled=1000, 1 MB records", False, 21, MAXH, off, b"dna_rolling")
