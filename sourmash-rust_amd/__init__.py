"""sourmash MinHash hot path on AMD MI355X (gfx950).

Host-side mirror of the reference's interface for the path (KmerMinHash, hash_murmur,
Signature) over the C ABI of include/sourmash.h; the compute is hand-written HIP in
csrc/ (libsourmash_amd.so).  There is no CPU fallback.
"""
from ._lib import SO_PATH, build, exported_symbols, lib  # noqa: F401
from .errors import SourmashError  # noqa: F401
from .minhash import KmerMinHash, hash_murmur, hash_words  # noqa: F401
from . import index, matrix  # noqa: F401


def device_available():
    return bool(lib().smh_device_available())
