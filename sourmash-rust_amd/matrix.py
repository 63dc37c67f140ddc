"""All-vs-all comparison (the N x N matrix the north star names): the reference has no matrix
entry point -- it is N^2 independent calls of KmerMinHash::compare (src/lib.rs:501-508), see
SURVEY.md 3.4.  Here: one device launch per block through the additive C ABI."""
import ctypes as C

import numpy as np

import contextlib

from ._lib import SmhCompareStats, SmhCompareTuning, f64p, lib, u64p
from .errors import call

ROUTES = {"auto": 0, "wave": 1, "few": 2, "components": 3, "tiled": 4}
DICTIONARIES = {"auto": 0, "full": 1}


@contextlib.contextmanager
def tuning(route="auto", visit_all_tiles=False, use_symmetry=True, comp_pairs_limit=96 << 10, split_frequent=True, dictionary="auto",
           range_masks=True):
    """Pins which kernel serves the block compares inside the `with` (additive ABI
    smh_compare_set_tuning; results never depend on it), then restores the defaults."""
    t = SmhCompareTuning(ROUTES[route], int(visit_all_tiles), int(use_symmetry), comp_pairs_limit, int(split_frequent),
                         DICTIONARIES[dictionary], 0 if range_masks else 1)
    call(lib().smh_compare_set_tuning, C.byref(t))
    try:
        yield
    finally:
        call(lib().smh_compare_set_tuning, None)


def last_stats():
    st = SmhCompareStats()
    lib().smh_compare_last_stats(C.byref(st))
    names = {v: k for k, v in ROUTES.items()}
    return {"route": names.get(st.route, "none"), "rows_per_tile": st.rows_per_tile, "tiles_visited": st.tiles_visited,
            "tiles_total": st.tiles_total, "pairs_per_tile": st.pairs_per_tile, "lds_overflow_steps": st.lds_overflow_steps,
            "frequent_hashes": st.frequent_hashes, "pipelined": st.pipelined,
            "span_halvings": st.span_halvings, "prefetched_after_halving": st.prefetched_after_halving}


def compare_block(rows, cols, want=("jaccard",)):
    """rows/cols: lists of KmerMinHash.  Returns dict name -> (len(rows), len(cols)) array.
    names: jaccard, common, size, count_common, containment."""
    n, m = len(rows), len(cols)
    out = {}
    bufs = {"jaccard": np.float64, "common": np.uint64, "size": np.uint64, "count_common": np.uint64,
            "containment": np.float64}
    for k in want:
        out[k] = np.zeros((n, m), dtype=bufs[k])

    def ptr(name):
        if name not in out:
            return None
        return out[name].ctypes.data_as(f64p if out[name].dtype == np.float64 else u64p)

    R = (C.c_void_p * max(n, 1))(*[r._p for r in rows])
    Cc = (C.c_void_p * max(m, 1))(*[c._p for c in cols])
    call(lib().smh_compare_block, R, n, Cc, m, ptr("jaccard"), ptr("common"), ptr("size"), ptr("count_common"),
         ptr("containment"))
    return out


def csr_from_sketches(sketches):
    """list of ascending uint64 arrays -> (flat uint64 array, uint64 offsets)."""
    off = np.zeros(len(sketches) + 1, dtype=np.uint64)
    for i, s in enumerate(sketches):
        off[i + 1] = off[i] + len(s)
    flat = np.concatenate([np.asarray(s, dtype=np.uint64) for s in sketches]) if sketches else np.zeros(0, np.uint64)
    return np.ascontiguousarray(flat), off


def compare_block_dev(row_hashes, row_offsets, col_hashes, col_offsets, num, want=("jaccard",), stream=None):
    """Device-resident CSR sketches (torch uint64/int64 CUDA tensors) -> dict of CUDA tensors.
    torch is only the allocator here; the work is the library's HIP kernel."""
    import torch
    n, m = len(row_offsets) - 1, len(col_offsets) - 1
    dev = row_hashes.device
    outs = {}
    for k in want:
        dt = torch.float64 if k in ("jaccard", "containment") else torch.int64
        outs[k] = torch.empty((n, m), dtype=dt, device=dev)

    def p(name):
        return C.c_void_p(outs[name].data_ptr()) if name in outs else C.c_void_p(0)

    ro = np.ascontiguousarray(row_offsets, dtype=np.uint64)
    co = np.ascontiguousarray(col_offsets, dtype=np.uint64)
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    call(lib().smh_compare_block_dev, C.c_void_p(row_hashes.data_ptr()), ro.ctypes.data_as(u64p), n,
         C.c_void_p(col_hashes.data_ptr()), co.ctypes.data_as(u64p), m, num,
         p("jaccard"), p("common"), p("size"), p("count_common"), p("containment"), C.c_void_p(stream))
    return outs


OWN_ALL, OWN_TRIANGLE, OWN_CIRCULAR = 0, 1, 2


class Collection:
    """The dictionary of one collection of sketches resident in HBM (additive ABI smh_collection_*): dense ranks of
    all hashes, components, frequent hashes -- built once, by this process alone (world=1) or together with the other
    ranks of a job (each sorts one slice of hash space; ONE all-gather of the shares in between), then any number of
    block compares.  hashes: CUDA int64/uint64 tensor (kept alive by this object); offsets: host uint64 array, n+1."""

    def __init__(self, hashes, offsets, world=1, rank=0, stream=None):
        import torch
        self._hashes = hashes
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = len(self.offsets) - 1
        self.world, self.rank = world, rank
        self._stream = stream if stream is not None else torch.cuda.current_stream(hashes.device).cuda_stream
        L = lib()
        self._p = None
        self._p = call(L.smh_collection_begin, C.c_void_p(hashes.data_ptr()), self.offsets.ctypes.data_as(u64p), self.n, world, rank,
                       C.c_void_p(self._stream))
        self.share_bytes = int(L.smh_collection_share_bytes(self._p))

    def share_to(self, dst):
        """copies this rank's share into `dst` (a CUDA uint8 tensor of share_bytes bytes, e.g. its slot of the gather buffer)"""
        assert dst.numel() * dst.element_size() == self.share_bytes and dst.is_contiguous()
        call(lib().smh_collection_share_to, self._p, C.c_void_p(dst.data_ptr()), C.c_void_p(self._stream))

    def finish(self, gathered=None):
        """gathered: CUDA uint8 tensor, world x share_bytes, rank-major (None when world == 1)"""
        if gathered is not None:
            assert gathered.is_contiguous() and gathered.numel() * gathered.element_size() == self.world * self.share_bytes
        self._gathered = gathered
        call(lib().smh_collection_finish, self._p, C.c_void_p(gathered.data_ptr() if gathered is not None else 0),
             C.c_void_p(self._stream))
        self._gathered = None

    def compare(self, row_lo, row_hi, num, want=("jaccard",), ownership=OWN_ALL):
        """rows [row_lo, row_hi) x all columns -> dict name -> CUDA tensor (row_hi - row_lo, n)"""
        import torch
        dev = self._hashes.device
        outs = {}
        for k in want:
            dt = torch.float64 if k in ("jaccard", "containment") else torch.int64
            outs[k] = torch.empty((row_hi - row_lo, self.n), dtype=dt, device=dev)

        def p(name):
            return C.c_void_p(outs[name].data_ptr()) if name in outs and outs[name].numel() else C.c_void_p(0)

        if row_hi > row_lo:
            call(lib().smh_collection_compare, self._p, row_lo, row_hi, num, ownership, p("jaccard"), p("common"), p("size"),
                 p("count_common"), p("containment"), C.c_void_p(self._stream))
        return outs

    def close(self):
        if self._p:
            lib().smh_collection_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
