"""C3: 1000 x 1000 num=2000 matrix on device-resident signatures, repeated; for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from sourmash_rust_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mode = sys.argv[2] if len(sys.argv) > 2 else "families"      # families | one_component | one_family | all_tiles | nosym | components
dense = mode in ("one_family", "components") or (len(sys.argv) > 4 and sys.argv[4] == "dense")
sigs = synth.family_signatures(0, n, num=2000, seed=3, n_families=1 if dense else 50)
if mode == "one_component":
    sigs[:, 0] = 1                                            # a contaminant hash shared by every signature
tune = {"all_tiles": dict(route="tiled", visit_all_tiles=True), "nosym": dict(route="tiled", visit_all_tiles=True, use_symmetry=False)}.get(mode, {})
t = torch.from_numpy(sigs.view(np.int64)).cuda()
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(2000)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if mode == "components":
    tune = dict(route="components")
if os.environ.get("PROF_FORCE_TILED"):
    tune = dict(tune, route="tiled")
if os.environ.get("PROF_DICT"):
    tune = dict(tune, dictionary=os.environ["PROF_DICT"])
if os.environ.get("PROF_NO_MASKS"):
    tune = dict(tune, range_masks=False)
if os.environ.get("PROF_COMP_LIMIT"):
    tune = dict(tune, comp_pairs_limit=int(os.environ["PROF_COMP_LIMIT"]))
import ctypes as C
L = pkg.lib()
times = []
with pkg.matrix.tuning(**tune):
    for it in range(iters):
        if it == 2:
            L.smh_profile_reset(); L.smh_profile_enable(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = pkg.matrix.compare_block_dev(t, off, t, off, 2000, want=("jaccard",))
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
L.smh_profile_enable(0)
kern = {}
for name in ("compare_tiled", "compare_comp", "compare_fill", "dictionary_rebuilt"):
    ms, k = C.c_double(), C.c_uint64()
    L.smh_profile_get(name.encode(), C.byref(ms), C.byref(k))
    if k.value:
        kern[name] = round(ms.value / k.value, 4) if name != "dictionary_rebuilt" else k.value
dt = sorted(times[2:] or times)[len(times[2:] or times) // 2]
print("n=%d %s: median %.3f ms per matrix (min %.3f; %.1f M pairs/s) kernels ms %s %s" % (n, mode, dt * 1e3, min(times) * 1e3, n * n / dt / 1e6, kern, pkg.matrix.last_stats()))
