"""PCIe-inclusive rate: the reference's boundary hands over HOST bytes.  2 GB of DNA in host memory
-> smh_add_sequence_len (one record) and smh_add_sequences (2000 records), k=31 scaled=1000."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
L = pkg.lib()
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000_000
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, None) == 0
host = buf.cpu().numpy()          # pageable host memory, like a caller's buffer
hb = host.ctypes.data_as(C.c_char_p)
MAXH = 18446744073709552
off = (np.arange(2001, dtype=np.uint64) * np.uint64(n // 2000)); off[-1] = n
u64p = C.POINTER(C.c_uint64)
for name in ("one record", "2000 records"):
    for it in range(3):
        mh = pkg.KmerMinHash(0, 31, False, 42, MAXH, False)
        t0 = time.perf_counter()
        if name == "one record":
            rc = L.smh_add_sequence_len(mh._p, hb, n, True)
        else:
            rc = L.smh_add_sequences(mh._p, hb, off.ctypes.data_as(u64p), 2000, True)
        sz = len(mh)
        dt = time.perf_counter() - t0
        assert rc == 0
    print("%-13s host bytes in: %.1f ms for %.1f GB = %.1f GB/s = %.1f G k-mers/s PCIe-inclusive (sketch %d)"
          % (name, dt * 1e3, n / 1e9, n / dt / 1e9, n / dt / 1e9, sz), flush=True)
