"""BASELINE config 5, one rank's share: protein arm (six-frame translation of DNA), ksize=27 (9
residues), scaled=1000, abundance tracking, 12.5 GB of DNA as 12 500 records x 1 MB resident in HBM.
Checks linearity at full size: sketch(all records) == merge(sketch(first half), sketch(second half))
(scaled mode: hash sets unite, abundances add).  python tools/bench_c5.py [n_records]"""
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
MAXH = 18446744073709552
def sketch(r0, r1):
    mh = pkg.KmerMinHash(0, 27, True, 42, MAXH, True)
    o = off[r0:r1 + 1] - off[r0]
    mh.add_sequences_dev(buf.data_ptr() + int(off[r0]), int(o[-1]), o, True)
    return mh
print("warm-up", flush=True)
sketch(0, min(nrec, 100))
torch.cuda.synchronize()
L.smh_profile_reset(); L.smh_profile_enable(1)
t0 = time.perf_counter(); whole = sketch(0, nrec); torch.cuda.synchronize(); dt_first = time.perf_counter() - t0
L.smh_profile_reset()
t0 = time.perf_counter(); whole = sketch(0, nrec); torch.cuda.synchronize(); dt = time.perf_counter() - t0
t0 = time.perf_counter(); n_whole = len(whole); dt_host = time.perf_counter() - t0
print("first call %.1f ms; steady state %.1f ms; bringing the sketch to the host %.1f ms"
      % (dt_first * 1e3, dt * 1e3, dt_host * 1e3))
L.smh_profile_enable(0)
windows = 0
for f in range(3):
    windows += 2 * max(0, (rlen - f) // 3 - 9 + 1) * nrec
print("C5 share: %d records x 1 MB, %.1f GB DNA -> %.2f G windows in %.1f ms = %.1f G windows/s (%.1f G bases/s), sketch %d hashes"
      % (nrec, total / 1e9, windows / 1e9, dt * 1e3, windows / dt / 1e9, total / dt / 1e9, n_whole), flush=True)
for name in (b"protein_fused", b"translate", b"hash_windows"):
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(name, C.byref(ms), C.byref(cnt))
    print("  kernel %s: %.2f ms over %d launches" % (name.decode(), ms.value, cnt.value))
h = nrec // 2
a, b = sketch(0, h), sketch(h, nrec)
a.merge(b)
wm, wa = whole.mins_np(), whole.abunds_np()
assert np.array_equal(a.mins_np(), wm), "linearity: mins differ"
assert np.array_equal(a.abunds_np(), wa), "linearity: abundances differ"
assert (wm[1:] > wm[:-1]).all() and wm[-1] <= MAXH
print("linearity at full size holds: |sketch| = %d, total abundance %d, ascending, all <= max_hash" % (wm.size, int(wa.sum())))
