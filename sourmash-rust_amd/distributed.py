"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU node, "gloo" in the CPU tests and in the shared-GPU rehearsals).  The reference has no distributed
code at all; what is sharded here is what the path offers.

Compare matrix (N x N, all-vs-all, row blocks stay on their rank).  Every pair is still exactly
KmerMinHash::compare of the two sketches (reference src/lib.rs:470-508); what is split is the work:

  1. ONE all-gather of the signatures: every rank holds every column.
  2. The dictionary pre-pass (pooled sort of all hashes -> dense ranks, components of the "shares a
     hash" graph, frequent hashes) is SHARDED by hash range: rank g sorts slice g of hash space
     (1/world of the pooled hashes), then ONE all-gather of the ranks' shares (a few bytes per hash)
     lets everybody assemble the whole dictionary (smh_collection_begin / _finish).
  3. Symmetry is kept: row i OWNS the pairs (i, j) with (j - i) mod N < N/2 (ties: i < j) -- every
     unordered pair has exactly one owner, every row owns N/2 pairs, so every rank walks 1/world of
     the upper triangle's work.  A rank computes the pairs its rows own (plus the mirrors inside its
     own diagonal block).
  4. ONE all-to-all hands every rank the transposed blocks its rows do NOT own: rank c sends
     out_c[:, rows of r]^T to rank r, which keeps the entries c's rows own.

Sketching shards records across ranks with no data-path collective; union_across_ranks() is the
optional final union of the per-rank partial sketches.

The compute is injected (`engine`) so that the CPU tests drive the same sharding, ownership and
exchange code with the oracle; the product default is the HIP path and raises without a GPU."""
import numpy as np


def shard_range(n_total, world, rank):
    """Contiguous, equal-size (ceil) row blocks; the last blocks may be short or empty."""
    per = (n_total + world - 1) // world
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi, per


# ---------------------------------------------------------------------------------------------
# pair ownership (the same rule as owns_pair() in csrc/compare_kernels.hip)

def owns(i, j, n):
    """Does row i own the pair (i, j) of an n x n all-vs-all matrix?  i, j: integer arrays/tensors
    (broadcastable).  (j - i) mod n < n/2, ties (n even, distance n/2) go to the smaller index."""
    d = (j - i) % n
    return (2 * d < n) | ((2 * d == n) & (i < j))


def block_needs(src_lo, src_hi, dst_lo, dst_hi, n):
    """Does any row of [src_lo, src_hi) own a pair with a row of [dst_lo, dst_hi)?  Decides whether
    src sends dst a block -- sender and receiver evaluate the same function.  (A superset test:
    distances (j - i) mod n of the two intervals form one circular interval.)"""
    if src_hi <= src_lo or dst_hi <= dst_lo:
        return False
    if (src_lo, src_hi) == (dst_lo, dst_hi):
        return False
    length = (src_hi - src_lo) + (dst_hi - dst_lo) - 1      # number of distinct differences j - i
    if length >= n:
        return True
    a = (dst_lo - (src_hi - 1)) % n                         # smallest difference, mod n
    half = n // 2                                           # distances 0..half may be owned
    return a <= half or a + length > n


# ---------------------------------------------------------------------------------------------
# transports

class _Comm:
    """all_gather / all_to_all over torch.distributed.  With gloo and CUDA tensors (the shared-GPU
    rehearsal) the data is staged through host memory."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.on = dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.stage = self.on and dist.get_backend(group) == "gloo"

    def all_gather(self, t):
        """t: contiguous tensor, same shape on every rank -> (world * t.shape[0], ...)"""
        import torch
        if self.world == 1:
            return t
        src = t.contiguous()
        if self.stage and src.is_cuda:
            h = src.cpu()
            out = torch.empty((self.world * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
            self.dist.all_gather_into_tensor(out, h, group=self.group)
            return out.to(t.device)
        out = torch.empty((self.world * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        self.dist.all_gather_into_tensor(out, src, group=self.group)
        return out

    def all_to_all(self, send, recv_numel, dtype, device):
        """send: list (per peer) of flat contiguous tensors (possibly empty); recv_numel: elements
        expected from every peer -> list of flat tensors."""
        import torch
        if self.world == 1:
            return [send[0]]
        dev = torch.device("cpu") if self.stage else device
        inp = torch.cat([s.reshape(-1).to(dev) for s in send]) if sum(s.numel() for s in send) else torch.empty(0, dtype=dtype, device=dev)
        out = torch.empty(sum(recv_numel), dtype=dtype, device=dev)
        self.dist.all_to_all_single(out, inp, list(recv_numel), [s.numel() for s in send], group=self.group)
        out = out.to(device)
        res, at = [], 0
        for k in recv_numel:
            res.append(out[at:at + k])
            at += k
        return res


# ---------------------------------------------------------------------------------------------
# engines: what computes a rank's row block

class HipEngine:
    """The product path: smh_collection_* (csrc/compare_kernels.hip)."""

    def begin(self, allsigs, n_total, world, rank):
        """allsigs: (>= n_total, width) tensor of fixed-width signatures, or a (flat hashes, host offsets) pair (ragged)"""
        from . import matrix
        if isinstance(allsigs, tuple):
            allsigs, offsets = allsigs
            self.offsets = np.ascontiguousarray(offsets[: n_total + 1], dtype=np.uint64)
        else:
            width = allsigs.shape[1]
            self.offsets = np.arange(n_total + 1, dtype=np.uint64) * np.uint64(width)
        self.coll = matrix.Collection(allsigs, self.offsets, world, rank)
        self.device = allsigs.device
        self.world, self.rank = world, rank

    def share(self):
        import torch
        t = torch.empty(self.coll.share_bytes, dtype=torch.uint8, device=self.device)
        self.coll.share_to(t)
        return t

    def finish(self, gathered):
        self.coll.finish(gathered)

    def compare(self, lo, hi, num, want, ownership):
        return self.coll.compare(lo, hi, num, want=want, ownership=ownership)

    def lengths(self, n_total):
        import torch
        return torch.from_numpy(np.diff(self.offsets).astype(np.int64)).to(self.device)

    def close(self):
        self.coll.close()


# ---------------------------------------------------------------------------------------------
# the exchange of the blocks a rank's rows do not own

def mirror_send_list(out, blocks, rank, n_total):
    """out: (n_local, n_total) tensor of this rank's row block.  -> per peer the flat transposed
    block out[:, rows of peer]^T when this rank's rows own pairs with the peer's rows, else empty.
    CUDA tensors: ONE call packs all the blocks into one buffer (smh_mirror_pack; the list holds views of it)."""
    lo, hi = blocks[rank]
    peers = [c for c, (clo, chi) in enumerate(blocks) if c != rank and block_needs(lo, hi, clo, chi, n_total)]
    if out.is_cuda and peers and hi > lo:
        import ctypes as C
        import torch
        from ._lib import lib
        from .errors import call
        assert out.is_contiguous() and out.element_size() == 8
        sizes = [(blocks[c][1] - blocks[c][0]) * (hi - lo) for c in peers]
        packed = torch.empty(sum(sizes), dtype=out.dtype, device=out.device)
        clo = (C.c_uint32 * len(peers))(*[blocks[c][0] for c in peers])
        chi = (C.c_uint32 * len(peers))(*[blocks[c][1] for c in peers])
        call(lib().smh_mirror_pack, C.c_void_p(out.data_ptr()), hi - lo, n_total, clo, chi, len(peers), C.c_void_p(packed.data_ptr()),
             C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        send, at = [out.new_empty(0) for _ in blocks], 0
        for c, k in zip(peers, sizes):
            send[c] = packed[at:at + k]
            at += k
        return send
    send = []
    for c, (clo, chi) in enumerate(blocks):
        if c in peers:
            send.append(out[:, clo:chi].t().contiguous().reshape(-1))
        else:
            send.append(out.new_empty(0))
    return send


def mirror_recv_sizes(blocks, rank, n_total):
    lo, hi = blocks[rank]
    return [(hi - lo) * (phi - plo) if p != rank and block_needs(plo, phi, lo, hi, n_total) else 0
            for p, (plo, phi) in enumerate(blocks)]


def mirror_apply(out, recv, blocks, rank, n_total):
    """keeps, from every received block, the entries the sender's rows own
    (CUDA tensors: smh_mirror_apply, one call for all blocks when they lie in one buffer)"""
    import torch
    lo, hi = blocks[rank]
    got = [p for p in range(len(blocks)) if p != rank and recv[p].numel()]
    if out.is_cuda and got:
        import ctypes as C
        from ._lib import lib
        from .errors import call
        flat = torch.cat([recv[p].reshape(-1) for p in got]) if len(got) > 1 and not _adjacent(recv, got) else recv[got[0]]
        plo = (C.c_uint32 * len(got))(*[blocks[p][0] for p in got])
        phi = (C.c_uint32 * len(got))(*[blocks[p][1] for p in got])
        call(lib().smh_mirror_apply, C.c_void_p(out.data_ptr()), lo, hi - lo, n_total, plo, phi, len(got), C.c_void_p(flat.data_ptr()),
             C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        return
    rows = torch.arange(lo, hi, device=out.device).unsqueeze(1)
    for p in got:
        plo, phi = blocks[p]
        cols = torch.arange(plo, phi, device=out.device).unsqueeze(0)
        theirs = owns(cols, rows, n_total)               # the sender's row j owns (j, i)
        blk = recv[p].reshape(hi - lo, phi - plo)
        out[:, plo:phi] = torch.where(theirs, blk, out[:, plo:phi])


def _adjacent(recv, got):
    """are the received blocks consecutive views of one buffer (what _Comm.all_to_all returns)?"""
    at = recv[got[0]].data_ptr()
    for p in got:
        if recv[p].data_ptr() != at or not recv[p].is_contiguous():
            return False
        at += recv[p].numel() * recv[p].element_size()
    return True


def _kernel_outputs(want, exchange):
    """containment = count_common / |row|: after a transpose the mirrored value would have the OTHER sketch's length as its
    denominator, so with an exchange it is derived afterwards from the (symmetric) count"""
    if not exchange:
        return tuple(want)
    extra = ("count_common",) if "containment" in want and "count_common" not in want else ()
    return tuple(k for k in want if k != "containment") + extra


def _ownership(world, symmetric):
    """smh_collection_compare's ownership: 2 = this rank's share of a matrix the ranks compute together, 1 = the whole
    matrix on one rank (upper triangle + mirrors), 0 = every pair of the block here"""
    if not symmetric:
        return 0
    return 2 if world > 1 else 1


def compare_matrix_sharded(local_sigs, n_total, num, want=("jaccard",), engine=None, group=None, symmetric=True,
                           timings=None):
    """local_sigs: (per, width) int64 tensor holding this rank's rows (rows beyond its share are
    padding), every rank with the same `per` = ceil(n_total / world).  Returns this rank's row
    block: dict name -> (n_local, n_total) tensor.  Collectives: all-gather of the signatures,
    all-gather of the dictionary shares, all-to-all of the mirrored blocks (see the module text);
    symmetric=False computes every pair of the row block locally instead (no all-to-all)."""
    import time
    import torch
    comm = _Comm(group)
    world, rank = comm.world, comm.rank
    blocks = [shard_range(n_total, world, r)[:2] for r in range(world)]
    lo, hi = blocks[rank]
    per = shard_range(n_total, world, rank)[2]
    assert local_sigs.shape[0] == per, "every rank passes ceil(n_total/world) rows (pad the last block)"
    eng = engine or HipEngine()
    want = tuple(want)
    mark = []

    def tick(name):
        if timings is not None:
            if local_sigs.is_cuda:
                torch.cuda.synchronize()
            mark.append((name, time.perf_counter()))

    tick("start")
    allsigs = comm.all_gather(local_sigs.contiguous())
    tick("all_gather_signatures")
    eng.begin(allsigs, n_total, world, rank)
    tick("dictionary_slice")
    gathered = comm.all_gather(eng.share()) if world > 1 else None
    tick("all_gather_shares")
    eng.finish(gathered)
    del gathered
    tick("dictionary_assemble")
    exchange = symmetric and world > 1
    kernel_want = _kernel_outputs(want, exchange)
    out = eng.compare(lo, hi, num, kernel_want, _ownership(world, symmetric))
    tick("compare")
    if exchange:
        for name in kernel_want:
            t = out[name]
            recv = comm.all_to_all(mirror_send_list(t, blocks, rank, n_total), mirror_recv_sizes(blocks, rank, n_total),
                                   t.dtype, t.device)
            mirror_apply(t, recv, blocks, rank, n_total)
        if "containment" in want:
            lens = eng.lengths(n_total)[lo:hi].to(torch.float64).unsqueeze(1)
            out["containment"] = out["count_common"].to(torch.float64) / lens
            if "count_common" not in want:
                del out["count_common"]
        tick("exchange_mirrors")
    eng.close()
    if timings is not None:
        for (_, t0), (name, t1) in zip(mark[:-1], mark[1:]):
            timings[name] = timings.get(name, 0.0) + (t1 - t0)
    return out


def sample_row_stretches(lo, hi, k_rows, seed=0):
    """Stretches [a, b) of a rank's rows [lo, hi) that verify_exchange recomputes: the first and the last rows of the block
    (where the ownership rule changes hands between neighbouring ranks) and stretches at seeded places in between;
    about k_rows rows in all, in pieces of at most 4."""
    n = hi - lo
    if n <= 0 or k_rows <= 0:
        return []
    if n <= k_rows:
        return [(lo, hi)]
    piece = max(1, min(4, k_rows // 4))
    starts = {lo, hi - piece}
    rng = np.random.RandomState(seed * 7919 + lo)
    while len(starts) * piece < k_rows:
        starts.add(lo + int(rng.randint(0, n - piece + 1)))
    return [(a, a + piece) for a in sorted(starts)]


def verify_exchange(local_sigs, n_total, num, out, names=None, k_rows=16, engine_factory=None, group=None, seed=0):
    """Is this rank's row block -- after the exchange of the mirrored blocks -- what the rank would have computed alone?
    Every rank all-gathers the signatures once more, builds the dictionary of the whole collection BY ITSELF (world 1: no
    shares, no slices), recomputes about `k_rows` sampled rows of its block with ownership 0 (every pair of the row walked
    here, nothing mirrored, nothing received) and compares them bit for bit with the same rows of `out`; the verdict is
    the conjunction over the ranks (all-reduce MIN).  Per pair the contract is KmerMinHash::compare / count_common
    (reference src/lib.rs:470-508, 428-436).  What a wrong mirror exchange, a wrong ownership rule or a wrong sliced
    dictionary would change, this catches on the first real RCCL run; it costs one extra dictionary per rank and is never
    inside a timed region.  -> {"ok": bool, "rows_checked": int (this rank), "names": [...]}"""
    import torch
    comm = _Comm(group)
    world, rank = comm.world, comm.rank
    lo, hi, per = shard_range(n_total, world, rank)
    names = [k for k in (names or out.keys()) if k != "containment" or "containment" in out]
    allsigs = comm.all_gather(local_sigs.contiguous())
    eng = (engine_factory or HipEngine)()
    eng.begin(allsigs, n_total, 1, 0)
    eng.finish(None)
    ok, rows = True, 0
    for a, b in sample_row_stretches(lo, hi, k_rows, seed):
        alone = eng.compare(a, b, num, tuple(names), 0)
        for k in names:
            mine = out[k][a - lo:b - lo]
            same = (alone[k] == mine) | ((alone[k] != alone[k]) & (mine != mine))      # NaN == NaN (containment of an empty row)
            ok = ok and bool(same.all().item())
        rows += b - a
    eng.close()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cpu" if comm.stage or not local_sigs.is_cuda else local_sigs.device)
    if comm.on and world > 1:
        comm.dist.all_reduce(flag, op=comm.dist.ReduceOp.MIN, group=group)
    return {"ok": bool(flag.item()), "rows_checked": rows, "names": list(names)}


def verify_union(mh, uni):
    """Does the united sketch `uni` hold every hash of this rank's partial sketch `mh`, with a count at least this rank's?
    Checked where both live, in HBM (searchsorted on the exported arrays; nothing is copied to the host but the verdict).
    KmerMinHash::merge of scaled sketches is a set union whose abundances add (reference src/lib.rs:307-403)."""
    import torch
    track = mh.track_abundance
    n_own, n_uni = mh.export_dev(), uni.export_dev()
    if n_own == 0:
        return True
    if n_uni < n_own:
        return False

    def arrays(s, n):
        m = torch.empty(n, dtype=torch.int64, device="cuda")
        a = torch.empty(n, dtype=torch.int64, device="cuda") if track else None
        s.export_dev(m, a)
        return m ^ (-1 << 63), a            # unsigned order as signed order

    om, oa = arrays(mh, n_own)
    um, ua = arrays(uni, n_uni)
    at = torch.searchsorted(um, om).clamp_(max=n_uni - 1)
    ok = bool((um[at] == om).all().item())
    if ok and track:
        ok = bool((ua[at] >= oa).all().item())
    return ok


def simulate_sharded(allsigs, n_total, num, world, want=("jaccard",), engine_factory=None, symmetric=True):
    """The same steps as compare_matrix_sharded for `world` ranks run one after the other in ONE
    process (no process group): the collectives become concatenations and list shuffles, everything
    else -- slices of the dictionary, ownership, the mirrored blocks -- is the code the ranks run.
    allsigs: (>= n_total, width) tensor, or (flat hashes, offsets) for ragged sketches.  Returns the list of the ranks' row blocks."""
    import torch
    blocks = [shard_range(n_total, world, r)[:2] for r in range(world)]
    engs = [(engine_factory or HipEngine)() for _ in range(world)]
    for r, e in enumerate(engs):
        e.begin(allsigs, n_total, world, r)
    gathered = torch.cat([e.share() for e in engs]) if world > 1 else None
    for e in engs:
        e.finish(gathered)
    want = tuple(want)
    exchange = symmetric and world > 1
    kernel_want = _kernel_outputs(want, exchange)
    outs = [e.compare(blocks[r][0], blocks[r][1], num, kernel_want, _ownership(world, symmetric)) for r, e in enumerate(engs)]
    if exchange:
        for name in kernel_want:
            sends = [mirror_send_list(outs[r][name], blocks, r, n_total) for r in range(world)]
            for r in range(world):
                sizes = mirror_recv_sizes(blocks, r, n_total)
                recv = [sends[p][r] for p in range(world)]
                assert [t.numel() for t in recv] == sizes, "sender and receiver disagree about a block"
                mirror_apply(outs[r][name], recv, blocks, r, n_total)
        if "containment" in want:
            for r, e in enumerate(engs):
                lens = e.lengths(n_total)[blocks[r][0]:blocks[r][1]].to(torch.float64).unsqueeze(1)
                outs[r]["containment"] = outs[r]["count_common"].to(torch.float64) / lens
                if "count_common" not in want:
                    del outs[r]["count_common"]
    for e in engs:
        e.close()
    return outs


def shard_records(n_records, world, rank):
    """Records [lo, hi) sketched by `rank` (no collective involved)."""
    lo, hi, _ = shard_range(n_records, world, rank)
    return lo, hi


def union_across_ranks(mh, group=None, parts=None):
    """The ranks' partial SCALED sketches of one input -> one sketch, on every rank, without leaving HBM: all-gather of the
    sizes, ONE all-gather of the padded hash arrays (and one of the abundances), then the parts are united on the device
    (smh_sketch_absorb_dev: rank arithmetic + scatters per part, no sort, no host copy).  The result is what ONE sketch
    fed all the ranks' records would hold -- the set union of the parts, abundances added (what add_hash does hash by hash,
    reference src/lib.rs:192-245; for two well-formed scaled sketches also what KmerMinHash::merge gives, src/lib.rs:307-403,
    except merge's quirk Q5: merging untracked sketches leaves `abunds = Some(...)`, the union here stays untracked).
    The parts must agree in what check_compatible compares (ksize, DNA/protein, max_hash, seed: src/lib.rs:176-190) and in
    track_abundance: the parameters travel with the sizes in the first (small) all-gather, and a mismatch raises the
    reference's Mismatch* error on EVERY rank before any data collective.
    `parts` (tests): a list of sketches standing in for the other ranks' (no process group needed)."""
    import torch
    from .errors import SourmashError
    from .minhash import KmerMinHash
    assert mh.num == 0 and mh.max_hash > 0, "the device union is for scaled sketches"
    comm = _Comm(group)
    track = mh.track_abundance
    locals_ = parts if parts is not None else [mh]

    def params(p):
        # what check_compatible looks at (reference src/lib.rs:176-190), + whether abundances are tracked (it decides
        # whether a rank takes part in the second all-gather: a mismatch there would hang the job, not fail it)
        mx = p.max_hash
        return [p.export_dev(), p.ksize, 1 if p.is_protein else 0, p.seed & 0xFFFFFFFF, p.seed >> 32,
                mx & 0xFFFFFFFF, mx >> 32, 1 if p.track_abundance else 0]

    table = [params(p) for p in locals_]
    if parts is None and comm.world > 1:
        t = torch.tensor(table, dtype=torch.int64, device="cuda")
        table = comm.all_gather(t).cpu().tolist()
    # every rank sees the same table, so every rank raises the same error -- before any data collective
    mine = params(mh)
    for r, row in enumerate(table):
        for col, code, what in ((1, 101, "ksize"), (2, 102, "DNA/protein"), (5, 103, "max_hash"), (6, 103, "max_hash"),
                                (3, 104, "seed"), (4, 104, "seed")):
            if row[col] != mine[col]:
                raise SourmashError(code, "union_across_ranks: part %d differs in %s" % (r, what))
        if row[7] != mine[7]:
            raise SourmashError(3, "union_across_ranks: part %d differs in track_abundance" % r)
    sizes = [int(row[0]) for row in table]
    cap = max(max(sizes), 1)

    def padded(p, want_ab):
        m = torch.zeros(cap, dtype=torch.int64, device="cuda")
        a = torch.zeros(cap, dtype=torch.int64, device="cuda") if want_ab else None
        p.export_dev(m, a)
        return m, a

    if parts is None:
        m, a = padded(mh, track)
        gm = comm.all_gather(m)
        ga = comm.all_gather(a) if track else None
    else:
        pairs = [padded(p, track) for p in parts]
        gm = torch.cat([x[0] for x in pairs])
        ga = torch.cat([x[1] for x in pairs]) if track else None
    out = KmerMinHash(0, mh.ksize, mh.is_protein, mh.seed, mh.max_hash, track)
    out.absorb_dev(gm, ga, [r * cap for r in range(len(sizes))], sizes)
    return out


def merge_sketch_across_ranks(mins, abunds=None, group=None):
    """Union of the ranks' partial sketches (ascending distinct uint64 `mins`, optional counts):
    all-gather of the lengths, then of the padded arrays; returns the concatenated per-rank lists
    [(mins_r, abunds_r)] on every rank for the caller to fold with KmerMinHash.merge (exact for
    scaled sketches: abundances add; for num sketches the mins are exact)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [(mins, abunds)]
    world = dist.get_world_size(group)
    dev = mins.device
    n = torch.tensor([mins.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)

    def gather(t):
        pad = torch.zeros(cap, dtype=torch.int64, device=dev)
        pad[: t.numel()] = t
        out = torch.empty(world * cap, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(out, pad, group=group)
        return [out[r * cap: r * cap + sizes[r]] for r in range(world)]

    gm = gather(mins)
    ga = gather(abunds) if abunds is not None else [None] * world
    return list(zip(gm, ga))
