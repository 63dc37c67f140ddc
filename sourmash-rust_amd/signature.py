"""Host-side mirror of the reference's Signature (src/lib.rs:546-675) over the C ABI
(src/ffi.rs:327-604)."""
import ctypes as C

from ._lib import lib
from .errors import call, take_str
from .minhash import KmerMinHash


class Signature:
    def __init__(self, _ptr=None):
        self._L = lib()
        self._p = _ptr if _ptr is not None else self._L.signature_new()

    def __del__(self):
        try:
            self._L.signature_free(self._p)
        except Exception:
            pass

    @property
    def name(self): return take_str(call(self._L.signature_get_name, self._p)).decode()
    @name.setter
    def name(self, v): call(self._L.signature_set_name, self._p, v.encode())
    @property
    def filename(self): return take_str(call(self._L.signature_get_filename, self._p)).decode()
    @filename.setter
    def filename(self, v): call(self._L.signature_set_filename, self._p, v.encode())
    @property
    def license(self): return take_str(call(self._L.signature_get_license, self._p)).decode()

    def push_mh(self, mh): call(self._L.signature_push_mh, self._p, mh._p)
    def set_mh(self, mh): call(self._L.signature_set_mh, self._p, mh._p)
    def first_mh(self): return KmerMinHash(0, 0, _ptr=call(self._L.signature_first_mh, self._p))

    def sketches(self):
        n = C.c_size_t()
        arr = call(self._L.signature_get_mhs, self._p, C.byref(n))
        return [KmerMinHash(0, 0, _ptr=arr[i]) for i in range(n.value)]

    def save_json(self): return take_str(call(self._L.signature_save_json, self._p)).decode()
    def __eq__(self, other): return bool(call(self._L.signature_eq, self._p, other._p))


def _wrap(arr, n):
    return [Signature(_ptr=arr[i]) for i in range(n)]


def load_signatures_buffer(data, ksize=0, moltype=None):
    """reference Signature::load_signatures via signatures_load_buffer (one Signature per sketch)."""
    n = C.c_size_t()
    arr = call(lib().signatures_load_buffer, data, len(data), False, ksize,
               moltype.encode() if moltype else None, C.byref(n))
    return _wrap(arr, n.value)


def load_signatures_path(path, ksize=0, moltype=None):
    n = C.c_size_t()
    arr = call(lib().signatures_load_path, path.encode(), False, ksize,
               moltype.encode() if moltype else None, C.byref(n))
    return _wrap(arr, n.value)


def from_json(data):
    """Signature::from_reader view (no flattening).  The reference ABI only exposes the flattened
    load, so each top-level element is loaded on its own and its sketches pushed back together."""
    import json
    docs = json.loads(data)
    if not isinstance(docs, list):
        load_signatures_buffer(data)  # let the library raise its serde error
    out = []
    for d in docs:
        flat = load_signatures_buffer(json.dumps([d]).encode())
        if not flat:
            out.append(Signature())
            continue
        s = flat[0]
        for extra in flat[1:]:
            s.push_mh(extra.first_mh())
        out.append(s)
    return out


def save_signatures(sigs):
    arr = (C.c_void_p * max(len(sigs), 1))(*[s._p for s in sigs])
    return take_str(call(lib().signatures_save_buffer, arr, len(sigs))).decode()
