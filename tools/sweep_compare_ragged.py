"""All-vs-all on ragged scaled-style sketches (sizes log-uniform in [300, 16000]) -- the shape of the
reference's .sbt.subset fixture scaled up."""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.RandomState(1)
sizes = np.exp(rng.uniform(np.log(300), np.log(16000), size=n)).astype(np.int64)
pool = np.sort(rng.randint(0, 1 << 62, size=400000, dtype=np.int64).astype(np.uint64))
pool = np.unique(pool)
sk = [np.sort(rng.choice(pool, s, replace=False)) for s in sizes]
flat, off = pkg.matrix.csr_from_sketches(sk)
t = torch.from_numpy(flat.view(np.int64)).cuda()
for want in (("jaccard",), ("jaccard", "containment")):
    out = pkg.matrix.compare_block_dev(t, off, t, off, 0, want=want)
    torch.cuda.synchronize()
    L.smh_profile_reset(); L.smh_profile_enable(1)
    t0 = time.perf_counter()
    out = pkg.matrix.compare_block_dev(t, off, t, off, 0, want=want)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, cnt = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"compare_tiled", C.byref(ms), C.byref(cnt))
    L.smh_profile_enable(0)
    j = out["jaccard"].cpu().numpy()
    i, k = 3, 77
    exp = len(np.intersect1d(sk[i], sk[k])) / len(np.union1d(sk[i], sk[k]))
    print("ragged N=%d (%d hashes) want=%s: total %.1f ms, tiled kernel %.1f ms, %.1f M pairs/s, check %s" % (
        n, flat.size, "+".join(want), dt * 1e3, ms.value / max(1, cnt.value), n * n / dt / 1e6, abs(j[i, k] - exp) < 1e-15))
