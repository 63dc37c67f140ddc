"""oracle/coracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of the C oracle (oracle/_build/liboracle.so, built by oracle/Makefile).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

u64p = C.POINTER(C.c_uint64)


def build(force=False):
    src = os.path.join(_HERE, "sourmash_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        build()
    L = C.CDLL(_SO)
    L.omh_hash_murmur.restype = C.c_uint64
    L.omh_hash_murmur.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
    L.omh_new.restype = C.c_void_p
    L.omh_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_int]
    L.omh_clone.restype = C.c_void_p
    L.omh_clone.argtypes = [C.c_void_p]
    L.omh_free.argtypes = [C.c_void_p]
    L.omh_add_hash.argtypes = [C.c_void_p, C.c_uint64]
    L.omh_add_word.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.omh_add_sequence.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_char_p, C.c_size_t]
    L.omh_add_many.argtypes = [C.c_void_p, u64p, C.c_size_t]
    L.omh_add_many_with_abund.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t]
    L.omh_add_from.argtypes = [C.c_void_p, C.c_void_p]
    L.omh_merge.argtypes = [C.c_void_p, C.c_void_p]
    L.omh_count_common.argtypes = [C.c_void_p, C.c_void_p, u64p]
    L.omh_intersection_size.argtypes = [C.c_void_p, C.c_void_p, u64p, u64p]
    L.omh_compare.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.omh_containment.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    L.omh_size.restype = C.c_size_t
    L.omh_size.argtypes = [C.c_void_p]
    L.omh_mins.restype = u64p
    L.omh_mins.argtypes = [C.c_void_p]
    L.omh_has_abunds.argtypes = [C.c_void_p]
    L.omh_abunds_size.restype = C.c_size_t
    L.omh_abunds_size.argtypes = [C.c_void_p]
    L.omh_abunds.restype = u64p
    L.omh_abunds.argtypes = [C.c_void_p]
    L.omh_mins_push.argtypes = [C.c_void_p, C.c_uint64]
    L.omh_abunds_push.argtypes = [C.c_void_p, C.c_uint64]
    L.omh_translate_frames.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p,
                                       C.POINTER(C.c_size_t)]
    L.omh_compare_matrix.argtypes = [u64p, u64p, C.c_size_t, u64p, u64p, C.c_size_t, C.c_uint32,
                                     C.c_uint32, C.c_uint64, u64p, u64p, C.POINTER(C.c_double)]
    L.osynth_splitmix64.restype = C.c_uint64
    L.osynth_splitmix64.argtypes = [C.c_uint64, C.c_uint64]
    L.osynth_dna.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
    _lib = L
    return L


class OracleError(Exception):
    def __init__(self, code, message=""):
        super().__init__("oracle status %d %s" % (code, message))
        self.code = code
        self.message = message


def hash_murmur(kmer, seed=42):
    kmer = bytes(kmer)
    return lib().omh_hash_murmur(kmer, len(kmer), seed)


class MinHash:
    """Thin handle over omh_t; mirrors the reference KmerMinHash methods."""

    def __init__(self, num, ksize, is_protein=False, seed=42, max_hash=0, track_abundance=False,
                 _ptr=None):
        self._L = lib()
        self._p = _ptr if _ptr is not None else self._L.omh_new(
            num, ksize, int(is_protein), seed, max_hash, int(track_abundance))
        self.num, self.ksize, self.is_protein = num, ksize, bool(is_protein)
        self.seed, self.max_hash = seed, max_hash

    def __del__(self):
        try:
            self._L.omh_free(self._p)
        except Exception:
            pass

    def copy(self):
        return MinHash(self.num, self.ksize, self.is_protein, self.seed, self.max_hash,
                       _ptr=self._L.omh_clone(self._p))

    @staticmethod
    def _chk(st, msg=""):
        if st != 0:
            raise OracleError(st, msg)

    def add_hash(self, h):
        self._chk(self._L.omh_add_hash(self._p, h))

    def add_word(self, w):
        w = bytes(w)
        self._chk(self._L.omh_add_word(self._p, w, len(w)))

    def add_many(self, hashes):
        import numpy as np
        a = np.ascontiguousarray(hashes, dtype=np.uint64)
        self._chk(self._L.omh_add_many(self._p, a.ctypes.data_as(u64p), a.size))

    def add_many_with_abund(self, items):
        import numpy as np
        items = list(items)
        h = np.ascontiguousarray([i[0] for i in items], dtype=np.uint64)
        a = np.ascontiguousarray([i[1] for i in items], dtype=np.uint64)
        self._chk(self._L.omh_add_many_with_abund(self._p, h.ctypes.data_as(u64p), a.ctypes.data_as(u64p), h.size))

    def add_sequence(self, seq, force=False):
        seq = bytes(seq)
        buf = C.create_string_buffer(max(64, self.ksize + 1))
        st = self._L.omh_add_sequence(self._p, seq, len(seq), int(force), buf, len(buf))
        # the offending k-mer is exactly ksize bytes and may hold NULs: take it by length
        self._chk(st, buf.raw[:self.ksize].decode("latin-1") if st == 1101 else "")

    def merge(self, other):
        self._chk(self._L.omh_merge(self._p, other._p))

    def add_from(self, other):
        self._chk(self._L.omh_add_from(self._p, other._p))

    def count_common(self, other):
        out = C.c_uint64()
        self._chk(self._L.omh_count_common(self._p, other._p, C.byref(out)))
        return out.value

    def intersection_size(self, other):
        c, s = C.c_uint64(), C.c_uint64()
        self._chk(self._L.omh_intersection_size(self._p, other._p, C.byref(c), C.byref(s)))
        return c.value, s.value

    def compare(self, other):
        out = C.c_double()
        self._chk(self._L.omh_compare(self._p, other._p, C.byref(out)))
        return out.value

    def containment(self, other):
        out = C.c_double()
        self._chk(self._L.omh_containment(self._p, other._p, C.byref(out)))
        return out.value

    @property
    def mins(self):
        n = self._L.omh_size(self._p)
        p = self._L.omh_mins(self._p)
        return [p[i] for i in range(n)]

    def mins_np(self):
        import numpy as np
        n = self._L.omh_size(self._p)
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(self._L.omh_mins(self._p), shape=(n,)).copy()

    @property
    def abunds(self):
        if not self._L.omh_has_abunds(self._p):
            return None
        n = self._L.omh_abunds_size(self._p)
        p = self._L.omh_abunds(self._p)
        return [p[i] for i in range(n)]

    def abunds_np(self):
        import numpy as np
        if not self._L.omh_has_abunds(self._p):
            return None
        n = self._L.omh_abunds_size(self._p)
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(self._L.omh_abunds(self._p), shape=(n,)).copy()

    def mins_push(self, v):
        self._L.omh_mins_push(self._p, v)

    def abunds_push(self, v):
        self._L.omh_abunds_push(self._p, v)


def synth_dna(start, length, seed, n_every=0):
    import numpy as np
    out = np.empty(length, dtype=np.uint8)
    lib().osynth_dna(out.ctypes.data, start, length, seed, n_every)
    return out


def compare_matrix(rows, cols, num, ksize=31, max_hash=0):
    """rows/cols: lists of ascending uint64 arrays.  Returns (common, size, jaccard) arrays."""
    import numpy as np

    def flat(sk):
        off = np.zeros(len(sk) + 1, dtype=np.uint64)
        for i, s in enumerate(sk):
            off[i + 1] = off[i] + len(s)
        data = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sk]) if sk else np.zeros(0, np.uint64)
        return np.ascontiguousarray(data), off

    rd, ro = flat(rows)
    cd, co = flat(cols)
    n, m = len(rows), len(cols)
    common = np.zeros((n, m), dtype=np.uint64)
    size = np.zeros((n, m), dtype=np.uint64)
    jac = np.zeros((n, m), dtype=np.float64)
    lib().omh_compare_matrix(rd.ctypes.data_as(u64p), ro.ctypes.data_as(u64p), n,
                             cd.ctypes.data_as(u64p), co.ctypes.data_as(u64p), m,
                             num, ksize, max_hash,
                             common.ctypes.data_as(u64p), size.ctypes.data_as(u64p),
                             jac.ctypes.data_as(C.POINTER(C.c_double)))
    return common, size, jac
