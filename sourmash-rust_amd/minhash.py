"""Host-side mirror of the reference's KmerMinHash (src/lib.rs:37-46, 141-513): same method
names, argument meaning and error behaviour, every call going through the C ABI."""
import ctypes as C

import numpy as np

from ._lib import lib, u64p
from .errors import call, check


def hash_murmur(kmer, seed=42):
    """reference src/lib.rs:33-35 via src/ffi.rs:15-24 (NUL-terminated input)."""
    return call(lib().hash_murmur, bytes(kmer), seed)


def hash_words(words, seed=42):
    """murmur64 of many byte strings in one device launch (additive ABI)."""
    words = [bytes(w) for w in words]
    off = np.zeros(len(words) + 1, dtype=np.uint64)
    for i, w in enumerate(words):
        off[i + 1] = off[i] + len(w)
    out = np.zeros(len(words), dtype=np.uint64)
    call(lib().smh_hash_words, b"".join(words), off.ctypes.data_as(u64p), len(words), seed,
         out.ctypes.data_as(u64p))
    return out


class KmerMinHash:
    def __init__(self, num, ksize, is_protein=False, seed=42, max_hash=0, track_abundance=False, _ptr=None):
        self._L = lib()
        self._p = _ptr if _ptr is not None else self._L.kmerminhash_new(
            num, ksize, bool(is_protein), seed, max_hash, bool(track_abundance))

    def __del__(self):
        try:
            self._L.kmerminhash_free(self._p)
        except Exception:
            pass

    # --- parameters (reference src/ffi.rs:190-242)
    @property
    def num(self): return self._L.kmerminhash_num(self._p)
    @property
    def ksize(self): return self._L.kmerminhash_ksize(self._p)
    @property
    def is_protein(self): return self._L.kmerminhash_is_protein(self._p)
    @property
    def seed(self): return self._L.kmerminhash_seed(self._p)
    @property
    def max_hash(self): return self._L.kmerminhash_max_hash(self._p)
    @property
    def track_abundance(self): return self._L.kmerminhash_track_abundance(self._p)

    # --- state
    def __len__(self): return self._L.kmerminhash_get_mins_size(self._p)

    def mins_np(self):
        n = len(self)
        p = call(self._L.kmerminhash_get_mins, self._p)
        out = np.ctypeslib.as_array(C.cast(p, u64p), shape=(max(n, 1),))[:n].copy()
        _free(p)
        return out

    @property
    def mins(self): return [int(x) for x in self.mins_np()]

    def abunds_np(self):
        if not self.track_abundance:
            return None
        n = self._L.kmerminhash_get_abunds_size(self._p)
        p = call(self._L.kmerminhash_get_abunds, self._p)
        out = np.ctypeslib.as_array(C.cast(p, u64p), shape=(max(n, 1),))[:n].copy()
        _free(p)
        return out

    @property
    def abunds(self):
        a = self.abunds_np()
        return None if a is None else [int(x) for x in a]

    def mins_push(self, v): self._L.kmerminhash_mins_push(self._p, v)
    def abunds_push(self, v): self._L.kmerminhash_abunds_push(self._p, v)

    # --- building (reference src/lib.rs:192-305, 405-426)
    def add_hash(self, h): call(self._L.kmerminhash_add_hash, self._p, h)
    def add_word(self, w): call(self._L.kmerminhash_add_word, self._p, bytes(w))

    def add_many(self, hashes):
        a = np.ascontiguousarray(hashes, dtype=np.uint64)
        call(self._L.smh_add_many, self._p, a.ctypes.data_as(u64p), a.size)

    def add_many_with_abund(self, items):
        """reference src/lib.rs:419-426: items = [(hash, abundance), ...]; each hash is added abundance times."""
        items = list(items)
        h = np.ascontiguousarray([i[0] for i in items], dtype=np.uint64)
        a = np.ascontiguousarray([i[1] for i in items], dtype=np.uint64)
        call(self._L.smh_add_many_with_abund, self._p, h.ctypes.data_as(u64p), a.ctypes.data_as(u64p), h.size)

    def add_sequence(self, seq, force=False):
        seq = bytes(seq)
        if b"\0" in seq:
            call(self._L.smh_add_sequence_len, self._p, seq, len(seq), bool(force))
        else:
            call(self._L.kmerminhash_add_sequence, self._p, seq, bool(force))

    def add_sequences(self, records, force=False):
        """Many records in one device pass (additive ABI smh_add_sequences)."""
        records = [bytes(r) for r in records]
        off = np.zeros(len(records) + 1, dtype=np.uint64)
        for i, r in enumerate(records):
            off[i + 1] = off[i] + len(r)
        call(self._L.smh_add_sequences, self._p, b"".join(records), off.ctypes.data_as(u64p), len(records), bool(force))

    def add_sequences_dev(self, dev_ptr, total_len, offsets, force=False, stream=None):
        """Records already resident in HBM: dev_ptr is a HIP device pointer (e.g. tensor.data_ptr())."""
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        call(self._L.smh_add_sequences_dev, self._p, C.c_void_p(dev_ptr), total_len, off.ctypes.data_as(u64p),
             off.size - 1, bool(force), C.c_void_p(stream or 0))

    @staticmethod
    def add_sequences_grouped(sketches, records, groups, force=False):
        """records[r] feeds sketches[groups[r]]: many signatures from one device pass
        (additive ABI smh_add_sequences_grouped)."""
        L = sketches[0]._L
        records = [bytes(r) for r in records]
        off = np.zeros(len(records) + 1, dtype=np.uint64)
        for i, r in enumerate(records):
            off[i + 1] = off[i] + len(r)
        grp = np.ascontiguousarray(groups, dtype=np.uint32)
        arr = (C.c_void_p * len(sketches))(*[m._p for m in sketches])
        call(L.smh_add_sequences_grouped, arr, len(sketches), b"".join(records), off.ctypes.data_as(u64p),
             grp.ctypes.data_as(C.POINTER(C.c_uint32)), len(records), bool(force))

    @staticmethod
    def add_sequences_grouped_dev(sketches, dev_ptr, total_len, offsets, groups, force=False, stream=None):
        L = sketches[0]._L
        off = np.ascontiguousarray(offsets, dtype=np.uint64)
        grp = np.ascontiguousarray(groups, dtype=np.uint32)
        arr = (C.c_void_p * len(sketches))(*[m._p for m in sketches])
        call(L.smh_add_sequences_grouped_dev, arr, len(sketches), C.c_void_p(dev_ptr), total_len, off.ctypes.data_as(u64p),
             grp.ctypes.data_as(C.POINTER(C.c_uint32)), off.size - 1, bool(force), C.c_void_p(stream or 0))

    # --- a scaled sketch's state as device arrays (additive ABI; the cross-rank union, distributed.union_across_ranks)
    def export_dev(self, mins_t=None, abunds_t=None, stream=None):
        """copies the ascending hashes (and abundances) into CUDA int64 tensors; returns the number of hashes
        (call without tensors to ask for the size)"""
        n = C.c_uint64()
        cap = mins_t.numel() if mins_t is not None else 0
        call(self._L.smh_sketch_export_dev, self._p, C.c_void_p(mins_t.data_ptr() if mins_t is not None else 0),
             C.c_void_p(abunds_t.data_ptr() if abunds_t is not None else 0), cap, C.byref(n), C.c_void_p(stream or 0))
        return n.value

    def absorb_dev(self, mins_t, abunds_t, part_starts, part_lens, stream=None):
        """unites this scaled sketch with sorted distinct parts lying in one CUDA buffer (see smh_sketch_absorb_dev)"""
        st = np.ascontiguousarray(part_starts, dtype=np.uint64)
        ln = np.ascontiguousarray(part_lens, dtype=np.uint64)
        call(self._L.smh_sketch_absorb_dev, self._p, C.c_void_p(mins_t.data_ptr()),
             C.c_void_p(abunds_t.data_ptr() if abunds_t is not None else 0), st.ctypes.data_as(u64p), ln.ctypes.data_as(u64p),
             st.size, C.c_void_p(stream or 0))

    def merge(self, other): call(self._L.kmerminhash_merge, self._p, other._p)
    def add_from(self, other): call(self._L.kmerminhash_add_from, self._p, other._p)

    # --- comparing (reference src/lib.rs:428-508, src/index.rs:146-154)
    def count_common(self, other): return call(self._L.kmerminhash_count_common, self._p, other._p)
    def compare(self, other): return call(self._L.kmerminhash_compare, self._p, other._p)
    def intersection(self, other): return call(self._L.kmerminhash_intersection, self._p, other._p)
    similarity = compare

    def check_compatible(self, other):
        """reference src/lib.rs:176-190: True, or SourmashError(101..104)."""
        call(self._L.smh_check_compatible, self._p, other._p)
        return True

    def intersection_hashes(self, other):
        """reference src/lib.rs:438-468 (Rust API): (common hashes, size of the combined sketch)."""
        p, n, sz = u64p(), C.c_uint64(), C.c_uint64()
        call(self._L.smh_intersection, self._p, other._p, C.byref(p), C.byref(n), C.byref(sz))
        out = [int(p[i]) for i in range(n.value)]
        _free(C.cast(p, C.c_void_p))
        return out, sz.value

    def intersection_size(self, other):
        c, s = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
        rows = (C.c_void_p * 1)(self._p)
        cols = (C.c_void_p * 1)(other._p)
        call(self._L.smh_compare_block, rows, 1, cols, 1, None, c.ctypes.data_as(u64p), s.ctypes.data_as(u64p), None, None)
        return int(c[0]), int(s[0])

    def containment(self, other):
        out = np.zeros(1, np.float64)
        rows = (C.c_void_p * 1)(self._p)
        cols = (C.c_void_p * 1)(other._p)
        call(self._L.smh_compare_block, rows, 1, cols, 1, None, None, None, None,
             out.ctypes.data_as(C.POINTER(C.c_double)))
        return float(out[0])


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _free(p):
    if p:
        _libc.free(p)
