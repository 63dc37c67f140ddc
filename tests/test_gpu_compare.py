"""GPU parity: compare / intersection_size / count_common / containment through the C ABI vs the
oracle and the committed golden matrices, bit-exact (the f64 results are quotients of exactly
representable integers: equality, not tolerance)."""
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

# Every oracle-compared matrix test runs over each way a block can be served (additive ABI
# smh_compare_set_tuning; the choice must never change a result): the shape-based default, the
# per-component pair kernel, the tiled matrix kernel over the tiles that can hold sharing pairs, and
# the tiled kernel over EVERY tile.  `expect` = the route smh_compare_last_stats must report.
ROUTES = [
    pytest.param(dict(), None, id="auto"),
    pytest.param(dict(route="components"), "components", id="components"),
    pytest.param(dict(route="tiled"), "tiled", id="tiled"),
    pytest.param(dict(route="tiled", visit_all_tiles=True), "tiled", id="tiled-all-tiles"),
    pytest.param(dict(route="tiled", visit_all_tiles=True, use_symmetry=False), "tiled", id="tiled-all-tiles-nosym"),
    pytest.param(dict(split_frequent=False), None, id="auto-no-frequent-split"),
    pytest.param(dict(route="tiled", split_frequent=False), "tiled", id="tiled-no-frequent-split"),
    # the pooled sort of the dictionary with all eight byte passes instead of four + the tie fix
    pytest.param(dict(dictionary="full"), None, id="auto-full-sort-dictionary"),
    pytest.param(dict(route="tiled", dictionary="full"), "tiled", id="tiled-full-sort-dictionary"),
    # the tiled kernel walking every pair from the first range on, instead of from where the range masks say its cut lies
    pytest.param(dict(route="tiled", range_masks=False), "tiled", id="tiled-no-range-masks"),
    pytest.param(dict(route="tiled", visit_all_tiles=True, range_masks=False), "tiled", id="tiled-all-tiles-no-range-masks"),
]


def routed(pkg, tune, expect, fn):
    with pkg.matrix.tuning(**tune):
        out = fn()
        st = pkg.matrix.last_stats()
    if expect is not None:
        assert st["route"] == expect, st
        if tune.get("visit_all_tiles") and not tune.get("use_symmetry", True):
            assert st["tiles_visited"] == st["tiles_total"], st
    return out


def mh_from_sketch(M, sk):
    mh = M(0 if sk["max_hash"] else sk["num"], sk["ksize"], sk["molecule"] == "protein", sk["seed"],
           sk["max_hash"], "abundances" in sk)
    for m in sk["mins"]:
        mh.mins_push(m)
    for a in sk.get("abundances", []):
        mh.abunds_push(a)
    return mh


def test_sbt_v5_hit_counts(pkg, sbt_v5_leaves):
    # reference src/index/sbt.rs:543-589 through kmerminhash_compare / containment
    mhs = {pos: mh_from_sketch(pkg.KmerMinHash, sk) for pos, sk in sbt_v5_leaves.items()}
    q = mhs[7]
    sims = [mh.compare(q) for mh in mhs.values()]
    cont = [mh.containment(q) for mh in mhs.values()]
    assert sum(v > 0.5 for v in sims) == 1 and sum(v > 0.1 for v in sims) == 2
    assert sum(v > 0.5 for v in cont) == 2 and sum(v > 0.1 for v in cont) == 4


@pytest.mark.parametrize("tune,expect", ROUTES)
@pytest.mark.parametrize("tag", ["v5", "subset"])
def test_golden_matrices(tag, tune, expect, pkg, sbt_v5_leaves, sbt_subset_sketches):
    mats = np.load(os.path.join(GOLDEN, "golden_matrices.npz"))
    sks = [sbt_v5_leaves[k] for k in sorted(sbt_v5_leaves)] if tag == "v5" else sbt_subset_sketches
    mhs = [mh_from_sketch(pkg.KmerMinHash, s) for s in sks]
    out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block(
        mhs, mhs, want=("jaccard", "common", "size", "count_common", "containment")))
    assert (out["common"] == mats[tag + "_common"]).all()
    assert (out["size"] == mats[tag + "_size"]).all()
    assert (out["jaccard"] == mats[tag + "_jaccard"]).all()
    assert (out["count_common"] == mats[tag + "_count_common"]).all()
    lens = np.array([len(s["mins"]) for s in sks], dtype=np.float64)
    assert (out["containment"] == mats[tag + "_count_common"].astype(np.float64) / lens[:, None]).all()


def test_random_pairs_all_modes(pkg, coracle):
    rng = random.Random(21)
    for trial in range(60):
        num_a = rng.choice([0, 1, 5, 20, 50, 2000])
        num_b = num_a if rng.random() < 0.6 or num_a == 0 else rng.choice([1, 7, 30])  # H6: nums may differ
        mx = 0 if num_a else 1 << 62
        universe = [rng.getrandbits(62) for _ in range(rng.choice([10, 80, 5000]))]
        ga, oa = pkg.KmerMinHash(num_a, 21, False, 42, mx), coracle.MinHash(num_a, 21, False, 42, mx)
        gb, ob = pkg.KmerMinHash(num_b, 21, False, 42, mx), coracle.MinHash(num_b, 21, False, 42, mx)
        for h in rng.choices(universe, k=rng.choice([0, 1, 30, 3000])):
            ga.add_hash(h); oa.add_hash(h)
        for h in rng.choices(universe, k=rng.choice([0, 1, 30, 3000])):
            gb.add_hash(h); ob.add_hash(h)
        assert ga.count_common(gb) == oa.count_common(ob)
        assert ga.intersection_size(gb) == oa.intersection_size(ob)
        assert gb.intersection_size(ga) == ob.intersection_size(oa)
        assert ga.compare(gb) == oa.compare(ob) and gb.compare(ga) == ob.compare(oa)
        assert ga.intersection(gb) == oa.intersection_size(ob)[1]
        c = ga.containment(gb)
        if len(oa.mins):
            assert c == oa.containment(ob)
        else:
            assert c != c  # 0/0 = NaN like the reference


def test_large_sketches_not_in_lds(pkg, coracle):
    rng = np.random.RandomState(3)
    pool = np.unique(rng.randint(0, 1 << 62, size=60000, dtype=np.int64).astype(np.uint64))
    a = np.sort(rng.choice(pool, 20000, replace=False))
    b = np.sort(rng.choice(pool, 25000, replace=False))
    ga, gb = pkg.KmerMinHash(0, 21, False, 42, 1 << 62), pkg.KmerMinHash(0, 21, False, 42, 1 << 62)
    ga.add_many(a); gb.add_many(b)
    exp = len(np.intersect1d(a, b))
    assert ga.count_common(gb) == exp
    assert ga.intersection_size(gb) == (exp, len(a) + len(b) - exp)


@pytest.mark.parametrize("tune,expect", ROUTES)
def test_device_csr_block(tune, expect, pkg, coracle):
    import torch
    rng = np.random.RandomState(5)
    pool = np.unique(rng.randint(0, 1 << 62, size=3000, dtype=np.int64).astype(np.uint64))
    rows = [np.sort(rng.choice(pool, rng.choice([0, 10, 500, 500, 500]), replace=False)) for _ in range(37)]
    cols = [np.sort(rng.choice(pool, rng.choice([1, 500, 500, 700]), replace=False)) for _ in range(53)]
    for num in (0, 500, 64):
        common, size, jac = coracle.compare_matrix(rows, cols, num, 31, 0 if num else 1 << 62)
        rf, ro = pkg.matrix.csr_from_sketches(rows)
        cf, co = pkg.matrix.csr_from_sketches(cols)
        rt = torch.from_numpy(rf.view(np.int64)).cuda()
        ct = torch.from_numpy(cf.view(np.int64)).cuda()
        out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block_dev(
            rt, ro, ct, co, num, want=("jaccard", "common", "size", "count_common")))
        torch.cuda.synchronize()
        assert (out["common"].cpu().numpy().view(np.uint64) == common).all()
        assert (out["size"].cpu().numpy().view(np.uint64) == size).all()
        assert (out["jaccard"].cpu().numpy() == jac).all()


def test_linear_index_find_and_scaffold(pkg, coracle, sbt_v5_leaves):
    # reference src/index/sbt.rs:567-601 on tests/data/v5.sbt.json
    leaves = {pos: mh_from_sketch(pkg.KmerMinHash, sk) for pos, sk in sbt_v5_leaves.items()}
    oleaves = {pos: mh_from_sketch(coracle.MinHash, sk) for pos, sk in sbt_v5_leaves.items()}
    lin = pkg.index.LinearIndex()
    order = sorted(leaves)
    for p in order:
        lin.insert(leaves[p])
    q = leaves[7]
    assert len(lin.find(pkg.index.search_minhashes, q, 0.5)) == 1
    assert len(lin.find(pkg.index.search_minhashes, q, 0.1)) == 2
    assert len(lin.find(pkg.index.search_minhashes_containment, q, 0.5)) == 2
    assert len(lin.find(pkg.index.search_minhashes_containment, q, 0.1)) == 4
    got = pkg.index.search_minhashes([leaves[p] for p in order], q, 0.05)
    exp = [i for i, p in enumerate(order) if oleaves[p].compare(oleaves[7]) > 0.05]
    assert got == exp
    # scaffold pairing: every leaf ends up in exactly one pair (7 leaves -> 4 pairs, one single)
    pairs = pkg.index.scaffold_pairs([leaves[p] for p in order])
    assert len(pairs) == 4 and sum(1 for a, b in pairs if b is None) == 1
    # nearest leaf = arg-max count_common, first on ties, like the reference loop
    rest = [leaves[p] for p in order if p != 7]
    orest = [oleaves[p] for p in order if p != 7]
    pos, cm = pkg.index.most_common(q, rest)
    ocs = [oleaves[7].count_common(o) for o in orest]
    assert cm == max(ocs) and pos == ocs.index(max(ocs))


@pytest.mark.parametrize("tune,expect", ROUTES)
def test_tiled_block_edge_cases(tune, expect, pkg, coracle):
    """The block kernels on shapes that are not multiples of the 16 x 64 tiles, with empty and
    ragged sketches, per-row nums that differ (H6: row i's num truncates pair (i, j)), sketches
    that hold a dense stretch of rank space and duplicate-heavy families."""
    rng = np.random.RandomState(11)
    pool = np.unique(rng.randint(0, 1 << 62, size=20000, dtype=np.int64).astype(np.uint64))
    dense = pool[:3000]                                  # one sketch holding a dense prefix of rank space
    sizes = [0, 1, 5, 300, 300, 300, 300, 1200, 3000]
    rows, cols = [], []
    for i in range(83):
        k = sizes[i % len(sizes)]
        rows.append(dense if k == 3000 else np.sort(rng.choice(pool, k, replace=False)))
    for j in range(131):
        k = sizes[(j * 5 + 3) % len(sizes)]
        cols.append(dense[:2500] if k == 3000 else np.sort(rng.choice(pool, k, replace=False)))
    nums = [0, 1, 7, 300, 5000]
    row_mh, orow = [], []
    for i, r in enumerate(rows):
        n = nums[i % len(nums)]
        g = pkg.KmerMinHash(n, 21, False, 42, 0); o = coracle.MinHash(n, 21, False, 42, 0)
        for h in r:
            g.mins_push(int(h)); o.mins_push(int(h))
        row_mh.append(g); orow.append(o)
    col_mh, ocol = [], []
    for c in cols:
        g = pkg.KmerMinHash(300, 21, False, 42, 0); o = coracle.MinHash(300, 21, False, 42, 0)
        for h in c:
            g.mins_push(int(h)); o.mins_push(int(h))
        col_mh.append(g); ocol.append(o)
    out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block(
        row_mh, col_mh, want=("jaccard", "common", "size", "count_common", "containment")))
    for i in range(len(rows)):
        for j in range(0, len(cols), 7):
            c, s_ = orow[i].intersection_size(ocol[j])
            assert (int(out["common"][i, j]), int(out["size"][i, j])) == (c, s_), (i, j)
            assert out["jaccard"][i, j] == orow[i].compare(ocol[j])
            assert int(out["count_common"][i, j]) == orow[i].count_common(ocol[j])
            if len(rows[i]):
                assert out["containment"][i, j] == orow[i].containment(ocol[j])
            else:
                assert np.isnan(out["containment"][i, j])
    # jaccard-only request takes the early-exit instantiation: same numbers
    out2 = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block(row_mh, col_mh, want=("jaccard", "common", "size")))
    assert (out2["jaccard"] == out["jaccard"]).all() and (out2["common"] == out["common"]).all()
    assert (out2["size"] == out["size"]).all()


@pytest.mark.parametrize("num", [0, 400])
def test_halved_span_ahead_of_a_prefetched_crossing(num, pkg, coracle):
    """The shape that made the pipelined tiled kernel fault on a development build of round 3 ("The fault of
    round 3"): sketches that are sparse over most of rank space -- the span of a stretch doubles while its segments fit the
    LDS stage, and the boundary crossings of the stretch after the next are requested ahead -- and then hold a burst of
    consecutive pool values, so that a stretch does NOT fit at the span it tried and is rebuilt with halved spans while
    crossings requested for the boundary it would have had are in flight.  Bursts at jittered places, so that the rebuild
    meets the prefetch at every alignment; rows and columns with bursts at the same and at different places.  Every pair
    against the oracle; the stats say that spans were halved and that prefetched crossings were used afterwards."""
    import torch
    rng = np.random.RandomState(5)
    pool = np.unique(rng.randint(0, 1 << 62, size=64000, dtype=np.int64).astype(np.uint64))[:60000]
    starts = np.arange(700, 59000, 1900) + rng.randint(0, 600, size=len(np.arange(700, 59000, 1900)))

    def sketch(k_sparse, burst_ids, every):
        idx = set(rng.choice(len(pool), k_sparse, replace=False).tolist())
        for b in burst_ids:
            idx.update(range(int(starts[b]), int(starts[b]) + 80 * every, every))
        return pool[np.array(sorted(idx))]

    rows, cols = [], []
    for i in range(70):
        kind = i % 5
        if kind == 0:
            rows.append(sketch(150, [], 1))                                   # sparse only: spans grow to 64 ranges
        elif kind == 4 and i == 4:
            rows.append(np.sort(rng.choice(pool, 4800, replace=False)))       # the longest sketch sets the range granularity
        else:
            rows.append(sketch(120 + 40 * kind, rng.choice(len(starts), 3 + kind, replace=False), 1 + kind % 3))
    for j in range(150):
        kind = j % 4
        cols.append(sketch(100 + 100 * kind, rng.choice(len(starts), 2 * kind, replace=False) if kind else [], 1 + j % 4))
    rf, ro = pkg.matrix.csr_from_sketches(rows)
    cf, co = pkg.matrix.csr_from_sketches(cols)
    tr, tc = torch.from_numpy(rf.view(np.int64)).cuda(), torch.from_numpy(cf.view(np.int64)).cuda()
    ocommon, osize, ojac = coracle.compare_matrix(rows, cols, num, 21, 0)
    occ = np.array([[len(np.intersect1d(a, b, assume_unique=True)) for b in cols] for a in rows], dtype=np.int64)
    halvings = after = 0
    # (range_masks=False: the walk from the first range on -- with the masks a pair is walked only around its cut, or, without
    # a cut, not at all; the last entry checks that route on the same input)
    for tune in (dict(route="tiled", range_masks=False), dict(route="tiled", visit_all_tiles=True, range_masks=False), dict(route="tiled")):
        with pkg.matrix.tuning(**tune):
            out = pkg.matrix.compare_block_dev(tr, ro, tc, co, num, want=("jaccard", "common", "size", "count_common"))
            st = pkg.matrix.last_stats()
        assert st["route"] == "tiled" and st["pipelined"] == 1, st
        if not tune.get("range_masks", True):
            halvings += st["span_halvings"]; after += st["prefetched_after_halving"]
        assert (out["jaccard"].cpu().numpy() == ojac).all()
        assert (out["common"].cpu().numpy().view(np.uint64) == ocommon).all()
        assert (out["size"].cpu().numpy().view(np.uint64) == osize).all()
        assert (out["count_common"].cpu().numpy() == occ).all()
    assert halvings > 50 and after > 50, (halvings, after)


def _profile_count(pkg, name):
    import ctypes as C
    ms, k = C.c_double(), C.c_uint64()
    pkg.lib().smh_profile_get(name, C.byref(ms), C.byref(k))
    return k.value


@pytest.mark.parametrize("num", [0, 150, 400])
def test_range_masks_with_many_words_per_range_and_cuts_near_a_sketch_end(num, pkg, coracle):
    """The corners of the masked tiled kernel ("Range masks"): (i) components whose sketches draw from a pool
    twenty times their length -- hundreds of shared hashes per range, so a range's words exceed the three the kernel keeps
    in registers (the tail loop); (ii) SHORT sketches next to long ones, cut by `num` close to their last element, so that
    the four-ranks-at-a-time windows of the cut-range walk run past the end of a sketch (sentinels by hand) and past the end
    of the rank array (the padding); (iii) two components in every 64-column tile plus unrelated sketches (per-component bits:
    the short forms must not be taken); (iv) the all-vs-all block with its self pairs.  Every pair against the oracle, with
    masks and without, symmetric and as a rows x columns block."""
    import torch
    rng = np.random.RandomState(11 + num)
    pools = [np.unique(rng.randint(0, 1 << 62, size=9000, dtype=np.int64).astype(np.uint64))[:8000] for _ in range(2)]
    sk = []
    for i in range(150):
        pool = pools[i % 2]
        if i % 7 == 3:
            ln = int(rng.randint(3, 40))                                  # short: its last range is its first
        elif i % 7 == 5:
            ln = int(rng.randint(num + 1, num + 6)) if num else 17        # the cut a step or two before the sketch's end
        else:
            ln = int(rng.randint(300, 420))
        sk.append(np.sort(rng.choice(pool, ln, replace=False)))
    for i in range(10):                                                   # sketches of their own (no component, no bit)
        sk.append(np.unique(rng.randint(0, 1 << 62, size=200, dtype=np.int64).astype(np.uint64)))
    order = rng.permutation(len(sk))
    sk = [sk[i] for i in order]
    # the last sketch of the collection ends the rank array: make it one whose cut is near its end
    sk.append(np.sort(rng.choice(pools[0], (num + 2) if num else 9, replace=False)))
    flat, off = pkg.matrix.csr_from_sketches(sk)
    t = torch.from_numpy(flat.view(np.int64)).cuda()
    ocommon, osize, ojac = coracle.compare_matrix(sk, sk, num, 21, 0)
    occ = np.array([[len(np.intersect1d(a, b, assume_unique=True)) for b in sk] for a in sk], dtype=np.int64)
    rows = list(range(5, 61))
    rflat, roff = pkg.matrix.csr_from_sketches([sk[i] for i in rows])
    tr = torch.from_numpy(rflat.view(np.int64)).cuda()
    for tune in (dict(route="tiled"), dict(route="tiled", range_masks=False), dict(route="tiled", use_symmetry=False),
                 dict(route="tiled", visit_all_tiles=True), dict(route="tiled", dictionary="full"), dict()):
        with pkg.matrix.tuning(**tune):
            out = pkg.matrix.compare_block_dev(t, off, t, off, num, want=("jaccard", "common", "size", "count_common"))
            blk = pkg.matrix.compare_block_dev(tr, roff, t, off, num, want=("jaccard", "common", "count_common"))
        assert (out["jaccard"].cpu().numpy() == ojac).all(), tune
        assert (out["common"].cpu().numpy().view(np.uint64) == ocommon).all(), tune
        assert (out["size"].cpu().numpy().view(np.uint64) == osize).all(), tune
        assert (out["count_common"].cpu().numpy() == occ).all(), tune
        assert (blk["jaccard"].cpu().numpy() == ojac[rows]).all(), tune
        assert (blk["common"].cpu().numpy().view(np.uint64) == ocommon[rows]).all(), tune
        assert (blk["count_common"].cpu().numpy() == occ[rows]).all(), tune


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_four_pass_dictionary_equals_the_full_sort_dictionary(seed, pkg, coracle):
    """The pooled hashes of the dictionary are sorted by the 32 most significant bits that vary, and the keys that tie there
    are put in order afterwards (k_tie_fix, k_tie_sort); on request, or when that gives up, by all eight byte passes.
    Ragged random collections -- empty sketches, one long sketch, hashes that are small integers (no high bits: the span
    comes from the data), hashes near 2^64, hashes held by most sketches (runs of equal keys that tie with other keys) --
    must give the same matrix both ways, on the default route and on the tiled one, and the oracle's on sampled rows."""
    import torch
    rng = np.random.RandomState(100 + seed)
    n = [37, 700, 2500][seed - 1]
    shift = [0, 20, 44][seed - 1]
    pool = np.unique(rng.randint(1, 1 << 20, size=30000, dtype=np.int64).astype(np.uint64)) << np.uint64(shift)
    if seed == 3:
        pool = np.concatenate([pool, np.uint64(0xFFFFFFFFFFFFFFFF) - np.arange(50, dtype=np.uint64)[::-1]])
    common = pool[rng.choice(len(pool), 5, replace=False)]
    sk = []
    for i in range(n):
        k = int(rng.choice([0, 1, 3, 40, 200, 600]))
        if i == 5:
            k = 4000
        h = rng.choice(pool, k, replace=False) if k else np.zeros(0, np.uint64)
        if i % 3 and k:
            h = np.concatenate([h, common])               # five hashes held by two thirds of the sketches
        sk.append(np.unique(h))
    flat, off = pkg.matrix.csr_from_sketches(sk)
    t = torch.from_numpy(flat.view(np.int64)).cuda()
    num = [0, 150, 25][seed - 1]
    names = ("jaccard", "common", "size", "count_common", "containment")
    outs = {}
    for key, tune in (("b", dict()), ("r", dict(dictionary="full")), ("bt", dict(route="tiled")), ("rt", dict(route="tiled", dictionary="full")),
                      ("bn", dict(split_frequent=False))):
        with pkg.matrix.tuning(**tune):
            outs[key] = pkg.matrix.compare_block_dev(t, off, t, off, num, want=names)
    for key in ("r", "bt", "rt", "bn"):
        for name in names:
            a, b = outs["b"][name], outs[key][name]
            assert bool(((a == b) | ((a != a) & (b != b))).all()), (key, name)
    rows = sorted(set(rng.choice(n, min(n, 12), replace=False).tolist() + [5]))
    ocommon, osize, ojac = coracle.compare_matrix([sk[i] for i in rows], sk, num, 21, 0)
    idx = torch.tensor(rows, device="cuda")
    assert (outs["b"]["jaccard"][idx].cpu().numpy() == ojac).all()
    assert (outs["b"]["common"][idx].cpu().numpy().view(np.uint64) == ocommon).all()
    assert (outs["b"]["size"][idx].cpu().numpy().view(np.uint64) == osize).all()


def test_a_tie_group_too_long_for_lds_sends_the_dictionary_to_the_full_sort(pkg, coracle):
    """3 000 sketches that all hold the same four neighbouring hashes: 12 000 keys that tie in the 32 sorted bits and arrive
    interleaved -- more than k_tie_sort's LDS sort takes.  The build raises its overflow flag, the block compare notices
    where it synchronises anyway, rebuilds the dictionary with all eight passes and runs again: same results as the oracle,
    and the rebuild is on record.  Sharded over two owners (a share must not leave a rank void) as well."""
    import torch
    from sourmash_rust_amd import distributed as D
    rng = np.random.RandomState(9)
    n = 3000
    shared = np.array([1 << 40, (1 << 40) + 1, (1 << 40) + 2, (1 << 40) + 5], dtype=np.uint64)
    pool = np.unique(rng.randint(1, 1 << 62, size=200000, dtype=np.int64).astype(np.uint64))
    sk = [np.unique(np.concatenate([shared, rng.choice(pool, int(rng.randint(5, 60)), replace=False)])) for _ in range(n)]
    flat, off = pkg.matrix.csr_from_sketches(sk)
    t = torch.from_numpy(flat.view(np.int64)).cuda()
    before = _profile_count(pkg, b"dictionary_rebuilt")
    out = pkg.matrix.compare_block_dev(t, off, t, off, 30, want=("jaccard", "common", "size"))
    assert _profile_count(pkg, b"dictionary_rebuilt") == before + 1
    rows = [0, 1, 1500, 2999]
    ocommon, osize, ojac = coracle.compare_matrix([sk[i] for i in rows], sk, 30, 21, 0)
    idx = torch.tensor(rows, device="cuda")
    assert (out["jaccard"][idx].cpu().numpy() == ojac).all()
    assert (out["common"][idx].cpu().numpy().view(np.uint64) == ocommon).all()
    assert (out["size"][idx].cpu().numpy().view(np.uint64) == osize).all()
    outs = D.simulate_sharded((t, off), n, 30, 2, want=("jaccard",))
    assert bool((torch.cat([o["jaccard"] for o in outs]) == out["jaccard"]).all())


def test_keys_that_differ_only_in_their_low_bits_are_sorted_in_lds(pkg, coracle):
    """The pooled sort of the dictionary looks at the 32 most significant bits that vary and puts the few keys that tie there
    in order afterwards (k_tie_fix) -- good for hashes.  These are not: a span of 2^63 with thousands of distinct keys packed
    into 2^13 values at its top.  They tie in the sorted bits and arrive out of order from their sketches: one group of
    ~8 000 keys, which k_tie_sort sorts in LDS (no rebuild).  Same numbers as the oracle."""
    import torch
    rng = np.random.RandomState(77)
    top = (np.uint64(1) << np.uint64(63)) + np.arange(8192, dtype=np.uint64)
    low = np.unique(rng.randint(1, 1 << 40, size=3000, dtype=np.int64).astype(np.uint64))
    n = 70
    sk = [np.unique(np.concatenate([rng.choice(top, int(rng.randint(20, 100)), replace=False), rng.choice(low, 30, replace=False)]))
          for _ in range(n)]                      # (at most 7 000 keys in the group: it fits the LDS sort)
    flat, off = pkg.matrix.csr_from_sketches(sk)
    t = torch.from_numpy(flat.view(np.int64)).cuda()
    before = _profile_count(pkg, b"dictionary_rebuilt")
    out = pkg.matrix.compare_block_dev(t, off, t, off, 60, want=("jaccard", "common", "size", "count_common"))
    assert _profile_count(pkg, b"dictionary_rebuilt") == before
    ocommon, osize, ojac = coracle.compare_matrix(sk, sk, 60, 21, 0)
    assert (out["jaccard"].cpu().numpy() == ojac).all()
    assert (out["common"].cpu().numpy().view(np.uint64) == ocommon).all()
    assert (out["size"].cpu().numpy().view(np.uint64) == osize).all()
    occ = np.array([[len(np.intersect1d(a, b, assume_unique=True)) for b in sk] for a in sk], dtype=np.int64)
    assert (out["count_common"].cpu().numpy() == occ).all()


def _oracle_rows(coracle, sigs, rows, num):
    """coracle.compare_matrix of the sampled rows against ALL columns (two merges + two intersections
    per pair, reference src/lib.rs:470-508), spread over the host cores by row."""
    from concurrent.futures import ThreadPoolExecutor      # ctypes releases the GIL inside the C call
    cols = [sigs[i] for i in range(sigs.shape[0])]
    with ThreadPoolExecutor(max_workers=min(16, len(rows))) as ex:
        res = list(ex.map(lambda r: coracle.compare_matrix([sigs[r]], cols, num, 31, 0), rows))
    return (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]), np.concatenate([r[2] for r in res]))


def _full_size_matrix_check(pkg, coracle, n, seed, sample_rows, tunes):
    """All-vs-all at a BASELINE size: symmetry (all nums equal), unit diagonal, size == num, and the
    sampled rows x ALL columns against the C oracle, for the family collection of SURVEY.md 8d and
    for the same collection with one hash shared by every signature (one connected component: the
    tiled kernel has to walk every tile)."""
    import torch
    from sourmash_rust_amd import synth
    base = synth.family_signatures(0, n, num=2000, seed=seed)
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(2000)
    for contaminated in (False, True):
        sigs = base.copy()
        if contaminated:
            sigs[:, 0] = 1
        t = torch.from_numpy(sigs.view(np.int64)).cuda()
        ocommon, osize, ojac = _oracle_rows(coracle, sigs, sample_rows, 2000)
        for tune, expect in tunes:
            out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block_dev(t, off, t, off, 2000, want=("jaccard", "common", "size")))
            st = pkg.matrix.last_stats()
            if contaminated and st["route"] == "tiled" and not tune.get("split_frequent", True):
                assert st["tiles_visited"] * 2 >= st["tiles_total"], st        # one component: (the upper half of) every tile
            if contaminated and tune.get("split_frequent", True):
                assert st["frequent_hashes"] == 1, st                          # the shared hash is set aside
            j = out["jaccard"]
            assert bool((j == j.T).all()) and bool((j.diagonal() == 1.0).all())
            assert bool((out["size"] == 2000).all())
            idx = torch.tensor(sample_rows, device="cuda")
            assert (j[idx].cpu().numpy() == ojac).all(), (contaminated, tune)
            assert (out["common"][idx].cpu().numpy().view(np.uint64) == ocommon).all(), (contaminated, tune)
            assert (osize == 2000).all()
            fam = torch.arange(n, device="cuda") % 50
            same = fam[:, None] == fam[None, :]
            assert float(j[same].min()) > 0.2 and float(j[~same].max()) < 0.01
            del out, j, same


def test_matrix_at_benchmark_size_vs_oracle(pkg, coracle):
    """C3 at full size (1000 x 1000, num=2000) through every block route; 40 sampled rows x all
    columns against the C oracle."""
    rows = sorted(set([0, 1, 49, 50, 51, 999] + list(range(7, 1000, 29))))
    tunes = [(dict(), None), (dict(route="components"), "components"), (dict(route="tiled"), "tiled"),
             (dict(route="tiled", visit_all_tiles=True), "tiled"), (dict(route="tiled", use_symmetry=False), "tiled"),
             (dict(route="tiled", split_frequent=False), "tiled")]
    _full_size_matrix_check(pkg, coracle, 1000, 3, rows, tunes)


def test_matrix_at_c4_size_vs_oracle(pkg, coracle):
    """C4 at full size on one GPU (10 000 x 10 000, num=2000): the shape-based default route (the
    tiled kernel: > 96 Ki sharing pairs) on the family collection and on the one-component
    collection, 32 sampled rows x all 10 000 columns against the C oracle."""
    rows = sorted(set([0, 1, 49, 50, 4999, 5000, 9999] + list(range(13, 10000, 400))))
    _full_size_matrix_check(pkg, coracle, 10000, 4, rows, [(dict(), "tiled"), (dict(split_frequent=False), "tiled")])


@pytest.mark.parametrize("n,n_fam,sym", [(700, 7, True), (1500, 50, True), (900, 13, False)])
def test_device_plan_matches_an_independent_tile_count(n, n_fam, sym, pkg):
    """The block compare plans on the device (slot order by connected component, tiles that can hold
    sharing pairs, rows per tile).  For a collection whose components are known by construction --
    interleaved families, one hash pool each -- the number of tiles it launched must equal the count
    worked out here with numpy from that structure alone (component id = smallest member, slots in
    (component, index) order, a tile is launched iff some row slot and some column slot of it share a
    component, tiles wholly below the diagonal are mirrored when rows == columns with one num)."""
    import torch
    from sourmash_rust_amd import synth
    sigs = synth.family_signatures(0, n, num=300, n_families=n_fam, pool=600, private=100, seed=11)
    t = torch.from_numpy(sigs.view(np.int64)).cuda()
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(300)
    if sym:
        rows_t, ro, row_idx = t, off, np.arange(n)
    else:
        row_idx = np.arange(0, n, 3)                       # a different row set: every third signature
        rows_t = t[torch.from_numpy(row_idx).cuda()].contiguous()
        ro = np.arange(len(row_idx) + 1, dtype=np.uint64) * np.uint64(300)
    with pkg.matrix.tuning(route="tiled"):
        out = pkg.matrix.compare_block_dev(rows_t, ro, t, off, 300, want=("jaccard",))
        st = pkg.matrix.last_stats()
    assert st["route"] == "tiled"
    fam_c = np.arange(n) % n_fam                            # family == component (pools are disjoint, members overlap)
    fam_r = row_idx % n_fam
    # component id = root of the union-find = smallest node id in the component; rows are nodes 0.., columns follow
    # (or ARE the rows when both sides are one list): the relative ORDER of components is all that matters here
    first_r = {f: int(np.flatnonzero(fam_r == f)[0]) for f in range(n_fam)}
    comp_key = {f: first_r[f] for f in range(n_fam)} if not sym else {f: f for f in range(n_fam)}
    rslots = sorted(range(len(row_idx)), key=lambda i: (comp_key[fam_r[i]], i))
    cslots = sorted(range(n), key=lambda j: (comp_key[fam_c[j]], j))
    tr, tc = st["rows_per_tile"], 64
    tiles_r, tiles_c = -(-len(rslots) // tr), -(-n // tc)
    assert st["tiles_total"] == tiles_r * tiles_c
    rf = np.array([fam_r[i] for i in rslots]); cf = np.array([fam_c[j] for j in cslots])
    count = 0
    for ti in range(tiles_r):
        fr = set(rf[ti * tr:(ti + 1) * tr].tolist())
        for tj in range(tiles_c):
            if sym and tj * tc + tc - 1 < ti * tr:
                continue
            if fr & set(cf[tj * tc:(tj + 1) * tc].tolist()):
                count += 1
    assert st["tiles_visited"] == count, (st, count)
    j = out["jaccard"].cpu().numpy()
    same = fam_r[:, None] == fam_c[None, :]
    assert (j[~same] == 0).all() and (j[same] > 0).all()


@pytest.mark.parametrize("n_contaminants,where", [(1, "low"), (3, "spread"), (64, "spread"), (70, "spread")])
@pytest.mark.parametrize("tune", [dict(), dict(route="components"), dict(route="tiled"), dict(split_frequent=False)],
                         ids=["auto", "components", "tiled", "no-split"])
def test_frequent_hashes_are_set_aside_exactly(n_contaminants, where, tune, pkg, coracle):
    """Hashes held by (nearly) every sketch -- a contaminant k-mer -- would glue unrelated sketches into one
    component.  The block compare sets up to 64 such hashes aside and decides pairs that share nothing else
    from per-sketch records (which frequent hashes, at which position); more than 64 and nothing is set
    aside.  Either way every output equals the oracle's: ragged sketches, per-row nums (bottom-num cuts
    that fall before, between and after the contaminants), num = 0, rows != columns, empty sketches."""
    rng = np.random.RandomState(100 + n_contaminants)
    n_fam = 6
    pools = [np.unique(rng.randint(1 << 20, 1 << 62, size=700, dtype=np.int64).astype(np.uint64)) for _ in range(n_fam)]
    if where == "low":
        cont = np.arange(1, n_contaminants + 1, dtype=np.uint64) * np.uint64(7)
    else:
        cont = np.unique(rng.randint(1, 1 << 62, size=n_contaminants, dtype=np.int64).astype(np.uint64))
        cont[0] = 3                                         # one below everything, the rest anywhere

    def make(count, shift, frac):
        out = []
        for i in range(count):
            if (i + shift) % 17 == 16:
                out.append(np.zeros(0, dtype=np.uint64))    # empty
                continue
            fam = (i * 5 + shift) % n_fam
            own = rng.choice(pools[fam], rng.choice([40, 300, 500]), replace=False)
            keep = cont[rng.random_sample(len(cont)) < frac]
            out.append(np.unique(np.concatenate([own, keep])))
        return out

    rows = make(70, 0, 0.95)
    cols = make(150, 3, 0.9)
    nums = [0, 5, 60, 300, 5000]
    gr, orr = zip(*[_pair(pkg, coracle, nums[i % 5], r) for i, r in enumerate(rows)])
    gc, oc = zip(*[_pair(pkg, coracle, 77, c) for c in cols])
    want = ("jaccard", "common", "size", "count_common", "containment")
    with pkg.matrix.tuning(**tune):
        out = pkg.matrix.compare_block(list(gr), list(gc), want=want)
        st = pkg.matrix.last_stats()
    if tune.get("split_frequent", True):
        # set aside when at most 64 of them pass the threshold (a quarter of the 220 sketches), none otherwise
        assert st["frequent_hashes"] == (len(cont) if len(cont) <= 64 else 0), st
        if len(cont) <= 64 and st["route"] == "tiled":
            assert st["tiles_visited"] < st["tiles_total"]          # unrelated families are not walked
    else:
        assert st["frequent_hashes"] == 0
    for i in range(len(rows)):
        for j in range(len(cols)):
            assert (int(out["common"][i, j]), int(out["size"][i, j])) == orr[i].intersection_size(oc[j]), (i, j)
            assert out["jaccard"][i, j] == orr[i].compare(oc[j])
            assert int(out["count_common"][i, j]) == orr[i].count_common(oc[j])
            if len(rows[i]):
                assert out["containment"][i, j] == orr[i].containment(oc[j])
            else:
                assert np.isnan(out["containment"][i, j])


def test_tiled_global_merge_branch(pkg, coracle):
    """A range whose segments do not fit the tiled kernel's LDS stage is merged straight from global
    memory.  Sketches that pack 2 000 hashes into a sliver of hash space (where everything else is
    sparse) produce such ranges; smh_compare_last_stats must report that the branch ran, and every
    pair must still match the oracle."""
    rng = np.random.RandomState(41)
    spread = lambda k: np.unique(rng.randint(0, 1 << 62, size=k, dtype=np.int64).astype(np.uint64))
    lo = np.uint64(1) << np.uint64(61)
    clustered = lambda k: np.unique(lo + rng.randint(0, 1 << 20, size=k, dtype=np.int64).astype(np.uint64))
    pool = clustered(2600)
    rows, cols = [], []
    for i in range(70):
        rows.append(np.sort(rng.choice(pool, 2000, replace=False)) if i % 5 == 0 else spread(rng.choice([300, 2000])))
    for j in range(150):
        cols.append(np.sort(rng.choice(pool, 1800, replace=False)) if j % 4 == 1 else spread(rng.choice([100, 2000])))
    for num in (2000, 0, 150):
        common, size, jac = coracle.compare_matrix(rows, cols, num, 31, 0 if num else 1 << 62)
        import torch
        rf, ro = pkg.matrix.csr_from_sketches(rows)
        cf, co = pkg.matrix.csr_from_sketches(cols)
        rt = torch.from_numpy(rf.view(np.int64)).cuda()
        ct = torch.from_numpy(cf.view(np.int64)).cuda()
        # (range_masks=False: every range walked -- with the masks a pair is only walked around its cut, which may miss the sliver)
        for tune in (dict(route="tiled", range_masks=False), dict(route="tiled", visit_all_tiles=True, range_masks=False), dict(route="tiled")):
            with pkg.matrix.tuning(**tune):
                out = pkg.matrix.compare_block_dev(rt, ro, ct, co, num, want=("jaccard", "common", "size", "count_common"))
                st = pkg.matrix.last_stats()
            assert st["route"] == "tiled" and (st["lds_overflow_steps"] > 0 or tune.get("range_masks", True)), st
            assert (out["common"].cpu().numpy().view(np.uint64) == common).all()
            assert (out["size"].cpu().numpy().view(np.uint64) == size).all()
            assert (out["jaccard"].cpu().numpy() == jac).all()
            cc = np.array([[len(np.intersect1d(r, c)) for c in cols] for r in rows], dtype=np.uint64)
            assert (out["count_common"].cpu().numpy().view(np.uint64) == cc).all()


def test_end_to_end_genomes_to_matrix(pkg, coracle):
    """Whole path as a caller would use it: sketch a set of related synthetic genomes through the
    C ABI (one add_sequence per contig), wrap them in Signatures, write/read the .sig JSON, run the
    all-vs-all block and the index-style find -- every number equal to the oracle's."""
    from sourmash_rust_amd import signature as S
    rng = random.Random(31)
    base = bytearray(coracle.synth_dna(0, 120000, 77, 0))
    genomes = []
    for gi in range(24):
        g = bytearray(base)
        for _ in range(gi * 40):                       # point mutations: more for later genomes
            p = rng.randrange(len(g)); g[p] = rng.choice(b"ACGT")
        cut = sorted(rng.sample(range(1000, len(g) - 1000), 5))
        contigs = [bytes(g[a:b]) for a, b in zip([0] + cut, cut + [len(g)])]
        genomes.append(contigs)
    gms, oms = [], []
    for contigs in genomes:
        gm, om = pkg.KmerMinHash(400, 31, False, 42, 0, True), coracle.MinHash(400, 31, False, 42, 0, True)
        for c in contigs:
            gm.add_sequence(c, False); om.add_sequence(c, False)
        gms.append(gm); oms.append(om)
    sigs = []
    for i, gm in enumerate(gms):
        sg = S.Signature(); sg.name = "genome%d" % i; sg.push_mh(gm); sigs.append(sg)
    text = S.save_signatures(sigs)
    back = S.load_signatures_buffer(text.encode(), ksize=31, moltype="DNA")
    assert len(back) == 24 and all(b == s_ for b, s_ in zip(back, sigs))
    loaded = [b.first_mh() for b in back]
    assert all(l.mins == o.mins and l.abunds == o.abunds for l, o in zip(loaded, oms))
    out = pkg.matrix.compare_block(loaded, loaded, want=("jaccard", "containment"))
    for i in range(24):
        for j in range(24):
            assert out["jaccard"][i, j] == oms[i].compare(oms[j])
            assert out["containment"][i, j] == oms[i].containment(oms[j])
    assert out["jaccard"][0, 1] > out["jaccard"][0, 23] > 0          # similarity decays with mutations
    hits = pkg.index.search_minhashes(loaded, loaded[0], 0.5)
    assert hits == [i for i in range(24) if oms[i].compare(oms[0]) > 0.5] and 0 in hits


def test_resident_index(pkg, coracle, sbt_subset_sketches):
    """The HBM-resident index answers find / most_common / block compare exactly like the per-call paths."""
    nodes = [mh_from_sketch(pkg.KmerMinHash, s) for s in sbt_subset_sketches]
    onodes = [mh_from_sketch(coracle.MinHash, s) for s in sbt_subset_sketches]
    idx = pkg.index.ResidentIndex(nodes)
    assert len(idx) == 100
    for qi in (0, 17, 99):
        for thr in (0.0, 0.01, 0.2):
            assert idx.find(nodes[qi], thr) == [i for i in range(100) if onodes[i].compare(onodes[qi]) > thr]
            assert idx.find(nodes[qi], thr, containment=True) == \
                [i for i in range(100) if onodes[i].containment(onodes[qi]) > thr]
            assert idx.find(nodes[qi], thr) == pkg.index.search_minhashes(nodes, nodes[qi], thr)
        occ = [onodes[qi].count_common(o) for o in onodes]
        assert idx.most_common(nodes[qi]) == (occ.index(max(occ)), max(occ))
    sub = pkg.index.ResidentIndex(nodes[:20])
    out = sub.compare(idx, want=("jaccard", "count_common"))
    ref = pkg.matrix.compare_block(nodes[:20], nodes, want=("jaccard", "count_common"))
    assert (out["jaccard"] == ref["jaccard"]).all() and (out["count_common"] == ref["count_common"]).all()
    # the index against ITSELF: its dictionary (ranks, components, frequent hashes) is built by the first call and reused by
    # the later ones, on every route and for every output
    full = pkg.matrix.compare_block(nodes, nodes, want=("jaccard", "common", "size", "count_common", "containment"))
    for tune in (dict(), dict(route="tiled"), dict(route="components"), dict(), dict(split_frequent=False), dict()):
        with pkg.matrix.tuning(**tune):
            own = idx.compare(idx, want=("jaccard", "common", "size", "count_common", "containment"))
        for k in full:
            assert (own[k] == full[k]).all() or (k == "containment" and np.array_equal(own[k], full[k], equal_nan=True)), (k, tune)
    bad = pkg.KmerMinHash(0, 31, False, 42, 9223372036854776)
    with pytest.raises(pkg.SourmashError) as ei:
        idx.find(bad, 0.1)
    assert ei.value.code == 101


def _pair(pkg, coracle, n, mins):
    g = pkg.KmerMinHash(n, 21, False, 42, 0); o = coracle.MinHash(n, 21, False, 42, 0)
    for h in mins:
        g.mins_push(int(h)); o.mins_push(int(h))
    return g, o


@pytest.mark.parametrize("few_is_row", [False, True])
def test_few_vs_many_kernel(few_is_row, pkg, coracle):
    """k_compare_few (a handful of sketches against many; the LinearIndex / scaffold shape) in both
    orientations: ragged and empty sketches, per-row nums (H6), identical sketches, a query too
    long for LDS, and the jaccard-only early-exit instantiation."""
    rng = np.random.RandomState(5 + few_is_row)
    pool = np.unique(rng.randint(0, 1 << 62, size=30000, dtype=np.int64).astype(np.uint64))
    sizes = [0, 1, 5, 64, 65, 300, 300, 1200, 3000]
    nums = [0, 1, 7, 64, 300, 5000]
    many = [np.sort(rng.choice(pool, sizes[i % len(sizes)], replace=False)) for i in range(150)]
    few = [np.sort(rng.choice(pool, k, replace=False)) for k in (0, 1, 300, 2999, 9000)]
    many[10] = few[2].copy(); many[11] = few[3].copy(); many[12] = few[4][:3000].copy()
    gm, om = zip(*[_pair(pkg, coracle, nums[i % len(nums)], m) for i, m in enumerate(many)])
    gf, of = zip(*[_pair(pkg, coracle, nums[(i + 2) % len(nums)], f) for i, f in enumerate(few)])
    R, C_, oR, oC = (gf, gm, of, om) if few_is_row else (gm, gf, om, of)
    want = ("jaccard", "common", "size", "count_common", "containment")
    out = pkg.matrix.compare_block(list(R), list(C_), want=want)
    for i in range(len(R)):
        for j in range(len(C_)):
            assert (int(out["common"][i, j]), int(out["size"][i, j])) == oR[i].intersection_size(oC[j]), (i, j)
            assert out["jaccard"][i, j] == oR[i].compare(oC[j])
            assert int(out["count_common"][i, j]) == oR[i].count_common(oC[j])
            if len(oR[i].mins):
                assert out["containment"][i, j] == oR[i].containment(oC[j])
            else:
                assert np.isnan(out["containment"][i, j])
    out2 = pkg.matrix.compare_block(list(R), list(C_), want=("jaccard", "common", "size"))
    for k in ("jaccard", "common", "size"):
        assert (out2[k] == out[k]).all()


def test_inputs_produced_on_the_default_stream_are_ordered(pkg):
    """A caller without streams of its own (torch's current stream is the legacy default one, handle 0 =
    'no stream given') may still have asynchronous work in flight that produces the inputs -- an
    all-gather it has just waited on, a non-blocking copy.  The library's own stream must run after it."""
    import torch
    from sourmash_rust_amd import synth
    n = 700
    sigs = synth.family_signatures(0, n, num=2000, seed=9)
    good = torch.from_numpy(sigs.view(np.int64)).pin_memory()
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(2000)
    ref = None
    for trial in range(3):
        dev = torch.zeros((n, 2000), dtype=torch.int64, device="cuda")
        busy = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
        for _ in range(4):
            busy.random_()                          # keeps the default stream busy ahead of the copy
        dev.copy_(good, non_blocking=True)          # asynchronous: pinned source
        out = pkg.matrix.compare_block_dev(dev, off, dev, off, 2000, want=("jaccard",))["jaccard"]
        torch.cuda.synchronize()
        assert bool((out.diagonal() == 1.0).all())
        if ref is None:
            ref = out.clone()
        assert bool((out == ref).all())


def test_pairwise_calls_see_every_mutation(pkg, coracle):
    """The pairwise entry points keep a device copy of each sketch between calls; it is validated
    against the host vector before every use, so a change through ANY route must show."""
    rng = random.Random(3)
    pool = [rng.getrandbits(60) for _ in range(6000)]
    for num, mx in ((500, 0), (0, 1 << 61), (20000, 0)):
        ga, oa = pkg.KmerMinHash(num, 21, False, 42, mx, True), coracle.MinHash(num, 21, False, 42, mx, True)
        gb, ob = pkg.KmerMinHash(num, 21, False, 42, mx, True), coracle.MinHash(num, 21, False, 42, mx, True)

        def check():
            assert ga.compare(gb) == oa.compare(ob)
            assert gb.compare(ga) == ob.compare(oa)
            assert ga.count_common(gb) == oa.count_common(ob)
            assert ga.intersection_size(gb) == oa.intersection_size(ob)
            assert ga.compare(ga) == oa.compare(oa)

        check()                                                   # both empty
        for h in pool[:3000]:
            ga.add_hash(h); oa.add_hash(h)
        check()
        for h in pool[1500:4500]:
            gb.add_hash(h); ob.add_hash(h)
        check()
        ga.add_hash(pool[5000]); oa.add_hash(pool[5000])          # one more hash
        check()
        gb.merge(ga); ob.merge(oa)                                # merge
        check()
        seq = bytes(rng.choice(b"ACGT") for _ in range(40000))
        ga.add_sequence(seq, True); oa.add_sequence(seq, True)    # device ingest
        check()
        big = max(ga.mins) + 12345
        if num == 0 and big <= mx:
            ga.mins_push(big); oa.mins_push(big)                  # raw ABI push (keeps the vector ascending)
            ga.abunds_push(1); oa.abunds_push(1)
            check()
        gc = pkg.KmerMinHash(num, 21, False, 42, mx, True)        # created, compared once, dropped
        gc.add_many(np.array(pool[:100], dtype=np.uint64))
        oc = coracle.MinHash(num, 21, False, 42, mx, True)
        for h in pool[:100]:
            oc.add_hash(h)
        assert gc.compare(ga) == oc.compare(oa)


@pytest.mark.parametrize("tune,expect", ROUTES)
def test_row_block_that_is_a_slice_of_the_columns(tune, expect, pkg, coracle):
    """One rank's row block passed as a VIEW of the gathered signature set: the tiled pre-pass encodes
    the columns only and takes the rows' ranks from the same array."""
    import torch
    rng = np.random.RandomState(17)
    pool = np.unique(rng.randint(0, 1 << 62, size=5000, dtype=np.int64).astype(np.uint64))
    n, width = 240, 300
    sigs = np.stack([np.sort(rng.choice(pool, width, replace=False)) for _ in range(n)])
    _row_block_views(pkg, coracle, sigs, tune, expect)
    # the same with two hashes held by every signature (set aside as frequent; the row view's records come
    # from the columns' elements)
    sigs2 = sigs.copy()
    sigs2[:, 0] = 5; sigs2[:, -1] = (1 << 63) - 1
    st = _row_block_views(pkg, coracle, sigs2, tune, expect)
    if tune.get("split_frequent", True):
        assert st["frequent_hashes"] == 2, st


def _row_block_views(pkg, coracle, sigs, tune, expect):
    import torch
    n, width = sigs.shape
    allt = torch.from_numpy(sigs.view(np.int64)).cuda()
    st = None
    for lo, hi in ((0, 80), (80, 160), (170, 240)):
        rows = allt[lo:hi]
        ro = np.arange(hi - lo + 1, dtype=np.uint64) * np.uint64(width)
        co = np.arange(n + 1, dtype=np.uint64) * np.uint64(width)
        out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block_dev(
            rows, ro, allt, co, width, want=("jaccard", "common", "size", "count_common")))
        torch.cuda.synchronize()
        common, size, jac = coracle.compare_matrix(list(sigs[lo:hi]), list(sigs), width, 31, 0)
        assert (out["common"].cpu().numpy().view(np.uint64) == common).all()
        assert (out["size"].cpu().numpy().view(np.uint64) == size).all()
        assert (out["jaccard"].cpu().numpy() == jac).all()
        assert (out["jaccard"].cpu().numpy()[np.arange(hi - lo), np.arange(lo, hi)] == 1.0).all()
        st = pkg.matrix.last_stats()
    return st


@pytest.mark.parametrize("tune,expect", ROUTES)
@pytest.mark.parametrize("same_set,uniform_num", [(True, False), (False, False), (True, True)])
def test_components_and_disjoint_pairs(same_set, uniform_num, tune, expect, pkg, coracle):
    """The tiled path visits only tiles that can hold sharing pairs (connected components of the
    'shares a hash' graph) and fills the rest as disjoint.  Interleaved families, singletons, empty
    sketches, per-row nums, rows != columns, every output, against the oracle pair by pair."""
    rng = np.random.RandomState(23 + same_set)
    n_fam = 7
    pools = [np.unique(rng.randint(0, 1 << 62, size=900, dtype=np.int64).astype(np.uint64)) for _ in range(n_fam)]

    def make(count, shift):
        out = []
        for i in range(count):
            kind = (i + shift) % 11
            if kind == 9:
                out.append(np.zeros(0, dtype=np.uint64))                              # empty
            elif kind == 10:
                out.append(np.unique(rng.randint(0, 1 << 62, size=200, dtype=np.int64).astype(np.uint64)))  # singleton
            else:
                fam = (i * 3 + shift) % n_fam                                           # families interleaved
                out.append(np.sort(rng.choice(pools[fam], rng.choice([50, 300, 600]), replace=False)))
        return out

    rows = make(150, 0)
    cols = rows if same_set else make(210, 4)
    # one num on every row + the same list on both axes: tiles below the diagonal are produced by mirrored writes
    nums = [150] * 4 if uniform_num else [0, 40, 300, 5000]
    gr, orr = zip(*[_pair(pkg, coracle, nums[i % 4], r) for i, r in enumerate(rows)])
    if same_set:
        gc, oc = gr, orr
    else:
        gc, oc = zip(*[_pair(pkg, coracle, 77, c) for c in cols])
    want = ("jaccard", "common", "size", "count_common", "containment")
    out = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block(list(gr), list(gc), want=want))
    for i in range(len(rows)):
        for j in range(len(cols)):
            assert (int(out["common"][i, j]), int(out["size"][i, j])) == orr[i].intersection_size(oc[j]), (i, j)
            assert out["jaccard"][i, j] == orr[i].compare(oc[j])
            assert int(out["count_common"][i, j]) == orr[i].count_common(oc[j])
            if len(rows[i]):
                assert out["containment"][i, j] == orr[i].containment(oc[j])
            else:
                assert np.isnan(out["containment"][i, j])
    out2 = routed(pkg, tune, expect, lambda: pkg.matrix.compare_block(list(gr), list(gc), want=("jaccard", "size")))
    assert (out2["jaccard"] == out["jaccard"]).all() and (out2["size"] == out["size"]).all()


@pytest.mark.parametrize("n,rows_per_tile,pipelined", [(320, 8, 1), (1000, 8, 1), (1700, 16, 1), (4600, 32, 1)])
def test_tile_shape_and_kernel_follow_the_block_size(n, rows_per_tile, pipelined, pkg, coracle):
    """ONE family (every pair has to be walked) at four sizes: the device plan picks 8-row tiles while fewer than ~820
    sixteen-row tiles hold sharing pairs, 16-row tiles up to ~2400, 32-row tiles beyond (8 waves x 1 / 2 / 4 rows of the
    software-pipelined kernel k_compare_tiled_pf).  Whatever the shape, sampled rows x all columns equal the C oracle, the
    matrix is symmetric and its diagonal is 1 (reference src/lib.rs:470-508)."""
    import torch
    from sourmash_rust_amd import synth
    num = 2000
    sigs = synth.family_signatures(0, n, num=num, seed=5, n_families=1)
    t = torch.from_numpy(sigs.view(np.int64)).cuda()
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(num)
    with pkg.matrix.tuning(route="tiled"):
        out = pkg.matrix.compare_block_dev(t, off, t, off, num, want=("jaccard", "common", "size", "count_common"))
    st = pkg.matrix.last_stats()
    assert (st["route"], st["rows_per_tile"], st["pipelined"]) == ("tiled", rows_per_tile, pipelined), st
    j = out["jaccard"]
    assert bool((j == j.T).all()) and bool((j.diagonal() == 1.0).all())
    rows = sorted(set([0, 1, 15, 16, n // 2, n - 17, n - 1] + list(range(5, n, max(1, n // 11)))))
    cols = [sigs[k] for k in range(n)]
    ocommon, osize, ojac = coracle.compare_matrix([sigs[i] for i in rows], cols, num, 31, 0)
    idx = torch.tensor(rows, device="cuda")
    assert (out["jaccard"][idx].cpu().numpy() == ojac).all()
    assert (out["common"][idx].cpu().numpy().view(np.uint64) == ocommon).all()
    assert (out["size"][idx].cpu().numpy().view(np.uint64) == osize).all()
    cc = out["count_common"][idx].cpu().numpy().view(np.uint64)
    for a, i in enumerate(rows[:4]):
        want = [len(np.intersect1d(sigs[i], sigs[k])) for k in range(0, n, max(1, n // 50))]
        assert list(cc[a, ::max(1, n // 50)]) == want


def test_tiled_route_on_random_mid_size_blocks(pkg):
    """tests/stress_tiled.py for 20 seconds with a fixed seed: random ragged collections of 1300-3600 sketches (one or two
    sets, a few families, random num) through the tiled route -- 8-, 16- and 32-row tiles of the pipelined kernel -- against
    the C oracle on sampled rows, symmetry included."""
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stress_tiled.py"), "20", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "stress ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
