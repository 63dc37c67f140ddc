import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
n = 500_000_000
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, C.c_void_p(0))
torch.cuda.synchronize()
off = np.array([0, n], dtype=np.uint64)
for it in range(3):
    mh = pkg.KmerMinHash(0, 27, True, 42, 18446744073709552, True)
    mh.add_sequences_dev(buf.data_ptr(), n, off, True)
print(len(mh))
