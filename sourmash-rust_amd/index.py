"""Callers of the hot path in the reference's index layer, rebuilt on the block-compare kernels:

  LinearIndex.find          reference src/index/linear.rs:25-45 + src/index/search.rs:3-9
  most_common (scaffold)    reference src/index/sbt.rs:361-370 (nearest leaf = arg-max count_common)

A "node" here is a KmerMinHash (the reference's Leaf wraps a Signature whose first sketch is used,
src/index.rs:108-161)."""
import ctypes as C

import numpy as np

from ._lib import lib, u64p
from .errors import call


def search_minhashes(nodes, query, threshold):
    """indices i with nodes[i].similarity(query) > threshold"""
    return _find(nodes, query, threshold, False)


def search_minhashes_containment(nodes, query, threshold):
    """indices i with nodes[i].containment(query) > threshold"""
    return _find(nodes, query, threshold, True)


def _find(nodes, query, threshold, containment):
    n = len(nodes)
    arr = (C.c_void_p * max(n, 1))(*[m._p for m in nodes])
    out = (C.c_uint32 * max(n, 1))()
    cnt = C.c_uint32()
    call(lib().smh_find, arr, n, query._p, float(threshold), bool(containment), out, C.byref(cnt))
    return [int(out[i]) for i in range(cnt.value)]


class ResidentIndex:
    """Sketches copied once into HBM (additive ABI smh_index_*): repeated queries upload only the query."""

    def __init__(self, nodes):
        self._L = lib()
        self.nodes = list(nodes)
        arr = (C.c_void_p * max(len(self.nodes), 1))(*[m._p for m in self.nodes])
        self._h = call(self._L.smh_index_new, arr, len(self.nodes))

    def __del__(self):
        try:
            self._L.smh_index_free(self._h)
        except Exception:
            pass

    def __len__(self):
        return self._L.smh_index_len(self._h)

    def drop_dictionary(self):
        """gives back the dictionary an all-vs-all compare of the index with itself cached (smh_index_drop_dictionary)"""
        self._L.smh_index_drop_dictionary(self._h)

    def find(self, query, threshold, containment=False):
        out = (C.c_uint32 * max(len(self.nodes), 1))()
        cnt = C.c_uint32()
        call(self._L.smh_index_find, self._h, query._p, float(threshold), bool(containment), out, C.byref(cnt))
        return [int(out[i]) for i in range(cnt.value)]

    def most_common(self, leaf):
        pos, cm = C.c_uint32(), C.c_uint64()
        call(self._L.smh_index_most_common, self._h, leaf._p, C.byref(pos), C.byref(cm))
        return pos.value, cm.value

    def compare(self, other, want=("jaccard",)):
        n, m = len(self), len(other)
        kinds = {"jaccard": np.float64, "common": np.uint64, "size": np.uint64, "count_common": np.uint64,
                 "containment": np.float64}
        out = {k: np.zeros((n, m), dtype=kinds[k]) for k in want}

        def p(name):
            if name not in out:
                return None
            return out[name].ctypes.data_as(C.POINTER(C.c_double) if out[name].dtype == np.float64 else u64p)

        call(self._L.smh_index_compare, self._h, other._h, p("jaccard"), p("common"), p("size"), p("count_common"),
             p("containment"))
        return out


class LinearIndex:
    """reference src/index/linear.rs: leaves + find(search_fn, query, threshold).  The leaves are
    mirrored in HBM on the first find after an insert."""

    def __init__(self):
        self.leaves = []
        self._resident = None

    def insert(self, mh):
        self.leaves.append(mh)
        self._resident = None

    def find(self, search_fn, query, threshold):
        if search_fn in (search_minhashes, search_minhashes_containment):
            if self._resident is None:
                self._resident = ResidentIndex(self.leaves)
            hits = self._resident.find(query, threshold, containment=search_fn is search_minhashes_containment)
        else:
            hits = search_fn(self.leaves, query, threshold)
        return [self.leaves[i] for i in hits]


def most_common(leaf, candidates):
    """(position, count_common) of the candidate sharing the most hashes with `leaf`."""
    n = len(candidates)
    arr = (C.c_void_p * max(n, 1))(*[m._p for m in candidates])
    pos, cm = C.c_uint32(), C.c_uint64()
    call(lib().smh_most_common, leaf._p, arr, n, C.byref(pos), C.byref(cm))
    return pos.value, cm.value


def scaffold_pairs(datasets):
    """The leaf-pairing pass of the reference's scaffold (src/index/sbt.rs:356-373): pop the last
    leaf, pair it with the remaining leaf sharing the most hashes, repeat.  Returns the pairs."""
    datasets = list(datasets)
    pairs = []
    while datasets:
        nxt = datasets.pop()
        if not datasets:
            pairs.append((nxt, None))
            break
        pos, _ = most_common(nxt, datasets)
        pairs.append((nxt, datasets.pop(pos)))
    return pairs
