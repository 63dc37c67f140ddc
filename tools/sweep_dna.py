"""Times the rolling DNA kernel alone (HIP events inside the library) for the launch geometry given
in SOURMASH_AMD_DNA_CFG="threads,logR,hb"; prints G k-mers/s.  Usage: python tools/sweep_dna.py [GB]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
L = pkg.lib()
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
n = int(gb * 1e9)
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, C.c_void_p(0))
torch.cuda.synchronize()
off = np.array([0, n], dtype=np.uint64)
for it in range(3):
    if it == 1:
        L.smh_profile_reset(); L.smh_profile_enable(1)
    mh = pkg.KmerMinHash(0, k, False, 42, 18446744073709552, False)
    mh.add_sequences_dev(buf.data_ptr(), n, off, True)
ms, cnt = C.c_double(), C.c_uint64()
L.smh_profile_get(b"dna_rolling", C.byref(ms), C.byref(cnt))
per = ms.value / max(1, cnt.value)
print("cfg=%s k=%d kernel %.3f ms for %.1f GB -> %.1f G k-mers/s (retained %d)" % (
    os.environ.get("SOURMASH_AMD_DNA_CFG", "default"), k, per, gb, n / per / 1e6, len(mh)))
