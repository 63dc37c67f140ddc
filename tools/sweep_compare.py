"""Times the all-vs-all compare matrix for several N (single GPU): total wall time per call (incl. the
rank-encoding pre-pass) and the tiled kernel alone (HIP events inside the library)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
from sourmash_rust_amd import synth  # noqa: E402

L = pkg.lib()
for n in [int(x) for x in (sys.argv[1:] or ["1000", "4000", "10000"])]:
    sigs = synth.family_signatures(0, n, num=2000, seed=3)
    t = torch.from_numpy(sigs.view(np.int64)).cuda()
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(2000)
    out = pkg.matrix.compare_block_dev(t, off, t, off, 2000, want=("jaccard",))
    torch.cuda.synchronize()
    L.smh_profile_reset(); L.smh_profile_enable(1)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out = pkg.matrix.compare_block_dev(t, off, t, off, 2000, want=("jaccard",))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"compare_tiled", C.byref(ms), C.byref(cnt))
    L.smh_profile_enable(0)
    k = ms.value / max(1, cnt.value)
    j = out["jaccard"]
    print("N=%d total %.2f ms (%.1f M pairs/s)  tiled kernel %.2f ms (%.1f M pairs/s)  mean J %.4f diag_ok %s" % (
        n, dt * 1e3, n * n / dt / 1e6, k, n * n / k / 1e3, float(j.mean()), bool((j.diagonal() == 1).all())))
    del out, t
