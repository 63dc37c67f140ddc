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


class LinearIndex:
    def __init__(self):
        self.leaves = []

    def insert(self, mh):
        self.leaves.append(mh)

    def find(self, search_fn, query, threshold):
        return [self.leaves[i] for i in search_fn(self.leaves, query, threshold)]


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
