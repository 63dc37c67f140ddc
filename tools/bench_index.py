"""Row N2: LinearIndex::find of one query against n nodes -- per-call upload (smh_find) vs the
HBM-resident index (smh_index_find).  Run on the GPU box: python tools/bench_index.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (maps torch's HIP runtime first)
from __graft_entry__ import load_package
pkg = load_package()
from sourmash_rust_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
num = 2000
sigs = synth.family_signatures(0, n, num=num, seed=3)
nodes = []
for r in sigs:
    m = pkg.KmerMinHash(num, 31, False, 42, 0)
    m.add_many(r)
    nodes.append(m)
q = nodes[7]
t0 = time.perf_counter(); a = pkg.index.search_minhashes(nodes, q, 0.1); t_first = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(5): a = pkg.index.search_minhashes(nodes, q, 0.1)
t_call = (time.perf_counter() - t0) / 5
t0 = time.perf_counter(); idx = pkg.index.ResidentIndex(nodes); t_build = time.perf_counter() - t0
b = idx.find(q, 0.1)
t0 = time.perf_counter()
for _ in range(20): b = idx.find(q, 0.1)
t_res = (time.perf_counter() - t0) / 20
import ctypes as C
L = pkg.lib()
L.smh_profile_reset(); L.smh_profile_enable(1)
for _ in range(10): idx.find(q, 0.1)
L.smh_profile_enable(0)
for name in (b"compare_few", b"compare_wave", b"compare_tiled"):
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(name, C.byref(ms), C.byref(cnt))
    if cnt.value: print("  kernel %s: %.3f ms avg over %d launches" % (name.decode(), ms.value / cnt.value, cnt.value))
assert a == b and len(a) > 0
print("n=%d num=%d hits=%d | smh_find %.2f ms/query | index build %.1f ms, smh_index_find %.3f ms/query (%.1f M nodes/s)"
      % (n, num, len(a), t_call * 1e3, t_build * 1e3, t_res * 1e3, n / t_res / 1e6))

# the resident index against itself: the first call builds its dictionary, the later ones reuse it
if n <= 20000:
    import numpy as _np
    t0 = time.perf_counter(); m1 = idx.compare(idx, want=("jaccard",)); t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(3): m2 = idx.compare(idx, want=("jaccard",))
    t_next = (time.perf_counter() - t0) / 3
    assert (m1["jaccard"] == m2["jaccard"]).all() and (_np.diag(m2["jaccard"]) == 1.0).all()
    print("index x index (%d^2, host outputs): first call %.1f ms, later calls %.1f ms (dictionary kept with the index)" % (n, t_first * 1e3, t_next * 1e3))
