import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from sourmash_rust_amd import synth
L = pkg.lib()
n = 10000
sigs = synth.family_signatures(0, n, num=2000, seed=3)
allt = torch.from_numpy(sigs.view(np.int64)).cuda()
rows = allt[:1250].contiguous()
ro = np.arange(1251, dtype=np.uint64) * np.uint64(2000)
co = np.arange(n + 1, dtype=np.uint64) * np.uint64(2000)
for it in range(3):
    L.smh_profile_reset(); L.smh_profile_enable(1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pkg.matrix.compare_block_dev(rows, ro, allt, co, 2000, want=("jaccard",))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"compare_tiled", C.byref(ms), C.byref(cnt))
print("one rank of 8: 1250 x 10000: total %.2f ms, tiled kernel %.2f ms" % (dt * 1e3, ms.value))
