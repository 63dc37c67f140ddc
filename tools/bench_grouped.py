"""N genomes -> N scaled signatures: one smh_add_sequences_grouped_dev call vs a loop of
smh_add_sequences_dev calls.  Run on the GPU box: python tools/bench_grouped.py [n_genomes] [genome_bytes]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
total = n * glen
buf = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
rc = L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 11, 0, None)
assert rc == 0
torch.cuda.synchronize()
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(glen)
grp = np.arange(n, dtype=np.uint32)
mx = (1 << 64) // 1000
num = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def fresh(): return [pkg.KmerMinHash(num, 31, False, 42, 0 if num else mx, False) for _ in range(n)]
for it in range(2):
    a = fresh()
    L.smh_profile_reset(); L.smh_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pkg.KmerMinHash.add_sequences_grouped_dev(a, buf.data_ptr(), total, off, grp, True)
    t_call = time.perf_counter() - t0
    sz = sum(len(m) for m in a)
    t_g = time.perf_counter() - t0
    L.smh_profile_enable(0)
ms, cnt = C.c_double(), C.c_uint64()
L.smh_profile_get(b"dna_rolling", C.byref(ms), C.byref(cnt))
print("grouped call %.1f ms (hash kernel %.1f ms over %d launches), size readback %.1f ms" % (t_call * 1e3, ms.value, cnt.value, (t_g - t_call) * 1e3))
for it in range(2):
    b = fresh()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        b[i].add_sequences_dev(buf.data_ptr() + i * glen, glen, np.array([0, glen], dtype=np.uint64), True)
    sz2 = sum(len(m) for m in b)
    t_l = time.perf_counter() - t0
assert sz == sz2 and all(x.mins == y.mins for x, y in zip(a[:50], b[:50]))
print("%d genomes x %.1f Mbp, k=31 %s: grouped %.1f ms (%.1f G k-mers/s) | per-sketch loop %.1f ms (%.1f G k-mers/s)"
      % (n, glen / 1e6, ("num=%d" % num) if num else "scaled=1000", t_g * 1e3, total / t_g / 1e9, t_l * 1e3, total / t_l / 1e9))
