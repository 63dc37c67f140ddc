"""Randomised GPU-vs-oracle parity fuzzer (run on the GPU box): python tools/fuzz_parity.py SECONDS [SEED]
Sketch side: random parameters (k 1..70, num / scaled / both / neither, abundance, protein), random
sequences of mixed composition (repeats, N runs, lowercase, arbitrary bytes), single calls and
multi-record batches, repeated adds on one object.  Compare side: random ragged sketch sets through
the block compare.  Stops at the first mismatch and prints a reproducer."""
import os
import random
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch  # noqa: E402,F401
import coracle  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
rng = random.Random(seed)
print("fuzz seed", seed)


def rand_seq(n):
    mode = rng.choice(["dna", "dna", "dna", "lowmix", "repeat", "nruns", "bytes", "polyA"])
    if mode == "polyA":
        return bytes([rng.choice(b"ACGT")]) * n
    if mode == "repeat":
        unit = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(1, 40)))
        return (unit * (n // len(unit) + 1))[:n]
    out = bytearray(rng.choice(b"ACGT") for _ in range(n))
    if mode == "lowmix":
        for i in range(n):
            if rng.random() < 0.3:
                out[i] |= 0x20
    if mode == "nruns":
        for _ in range(rng.randint(0, 5)):
            p = rng.randrange(max(1, n)); ln = rng.randint(1, 60)
            out[p:p + ln] = b"N" * min(ln, n - p)
    if mode == "bytes":
        for _ in range(rng.randint(1, 6)):
            if n:
                out[rng.randrange(n)] = rng.choice([0, 1, 0x20, 0x2a, 0x4e, 0x7f, 0x80, 0xc3, 0xa9, 0xe2, 0x82, 0xac, 0xff])
    return bytes(out)


def gmsg(e):   # the offending k-mer as bytes (the library's message is UTF-8 like the reference's String)
    return e.message.split(": ", 1)[-1].encode("utf-8") if e.code == 1101 else b""


def omsg(e):   # the oracle binding hands the raw bytes back as latin-1
    return e.message.encode("latin-1") if e.code == 1101 else b""


def state(m):
    return (m.mins, m.abunds)


def run_both(g, o, fn):
    eg = eo = None
    try:
        fn(g)
    except pkg.SourmashError as e:
        eg = (e.code, gmsg(e))
    try:
        fn(o)
    except coracle.OracleError as e:
        eo = (e.code, omsg(e))
    return eg, eo


t_end = time.time() + budget
n_sk = n_cmp = 0
t_print = time.time()
while time.time() < t_end:
    if time.time() - t_print > 30:
        t_print = time.time()
        print("... %d sketch cases, %d compare blocks so far" % (n_sk, n_cmp), flush=True)
    if rng.random() < 0.12:
        # grouped sketching: record r feeds sketch groups[r]; each sketch must equal the oracle fed its records
        ng = rng.randint(2, 12)
        k = rng.choice([4, 9, 16, 21, 31, 32, 33, 51, 70])
        kind = rng.choice(["scaled", "scaled", "num", "num", "numtrack", "numtrack", "mixed", "protein"])
        seedv = rng.choice([42, 7])
        def params(gi):
            if kind == "scaled":
                return (0, k, False, seedv, 1 << 60, gi % 2 == 0)
            if kind == "protein":
                return (0, max(k, 6), True, seedv, 1 << 61, gi % 3 == 0)
            if kind == "numtrack":
                return (rng.choice([1, 5, 50, 300]), k, False, seedv, 0, gi % 3 != 0)
            if kind == "num":
                return (rng.choice([1, 5, 50, 300]), k, False, seedv, 0, False)
            return rng.choice([(0, k, False, seedv, 1 << 60, True), (20, k, False, seedv, 0, True), (20, k, False, seedv, 0, False),
                               (5, k, False, seedv, 1 << 62, False), (0, max(k, 6), True, seedv, 1 << 61, False)])
        cases = [params(gi) for gi in range(ng)]
        gs, os2 = [pkg.KmerMinHash(*c) for c in cases], [coracle.MinHash(*c) for c in cases]
        for _ in range(rng.randint(1, 2)):
            force = rng.random() < 0.7
            recs = [rand_seq(rng.choice([0, 1, k - 1, k, k + 1, 50, 151, 400, 400, 3000, 40000])) for _ in range(rng.randint(2, 60))]
            if rng.random() < 0.3:
                recs[rng.randrange(len(recs))] = rand_seq(rng.choice([8, 50])) * 800     # repetitive: few distinct k-mers
            groups = [rng.randrange(ng) for _ in recs]
            if rng.random() < 0.5:
                groups.sort()
            eg = eo = None
            try:
                pkg.KmerMinHash.add_sequences_grouped(gs, recs, groups, force)
            except pkg.SourmashError as e:
                eg = (e.code, gmsg(e))
            for r, gi in zip(recs, groups):
                try:
                    os2[gi].add_sequence(r, force)
                except coracle.OracleError as e:
                    if eo is None:
                        eo = (e.code, omsg(e))
            if eg != eo or any(state(a) != state(b) for a, b in zip(gs, os2)):
                print("GROUPED MISMATCH kind", kind, "k", k, "cases", cases, "errors", eg, eo)
                import pickle
                pickle.dump((cases, recs, groups, force), open("gpurun_out/fuzz_fail_grouped.pkl", "wb"))
                sys.exit(1)
        n_sk += 1
    elif rng.random() < 0.72:
        prot = rng.random() < 0.25
        k = rng.choice([1, 2, 3, 4, 5, 7, 9, 11, 15, 16, 17, 20, 21, 24, 25, 27, 30, 31, 32, 33, 40, 48, 51, 63, 64, 65, 70, 96, 127, 128, 129, 150])
        if prot and k < 3:
            k = 3
        style = rng.choice(["num", "num", "scaled", "scaled", "both", "neither"])
        num = rng.choice([1, 2, 5, 20, 100, 500]) if style in ("num", "both") else 0
        mx = rng.choice([1 << 63, 1 << 61, 1 << 58, (1 << 64) // 1000]) if style in ("scaled", "both") else 0
        case = (num, k, prot, rng.choice([42, 42, 7, (1 << 40) + 3]), mx, rng.random() < 0.6)
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        steps = []
        for _ in range(rng.randint(1, 3)):
            force = rng.random() < 0.6
            if rng.random() < 0.3:
                recs = [rand_seq(rng.choice([0, 1, k - 1 if k > 1 else 1, k, k + 1, 50, 151, 400])) for _ in range(rng.randint(1, 40))]
                steps.append(("batch", recs, force))
                eg = None
                try:
                    g.add_sequences(recs, force)
                except pkg.SourmashError as e:
                    eg = (e.code, gmsg(e))
                eo = None
                for r in recs:
                    try:
                        o.add_sequence(r, force)
                    except coracle.OracleError as e:
                        if eo is None:
                            eo = (e.code, omsg(e))
            else:
                n = rng.choice([0, 1, k, k + 3, 100, 1000, 5000, 70000, 300000])
                if style == "neither" or style == "both":
                    n = min(n, 5000)
                s_ = rand_seq(n)
                steps.append(("single", s_, force))
                eg, eo = run_both(g, o, lambda m: m.add_sequence(s_, force))
            if eg != eo or state(g) != state(o):
                print("SKETCH MISMATCH case", case, "errors", eg, eo)
                for st in steps:
                    print("  step", st[0], "force", st[2], "len", [len(x) for x in st[1]] if st[0] == "batch" else len(st[1]))
                import pickle
                pickle.dump((case, steps), open("gpurun_out/fuzz_fail.pkl", "wb"))
                sys.exit(1)
        # pairwise entry points (mirrored device copies): against itself, against a partner, after a merge
        g2, o2 = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        s2_ = rand_seq(rng.choice([k, 200, 3000]))
        run_both(g2, o2, lambda m: m.add_sequence(s2_, True))
        for step in range(2):
            got = (g.compare(g), g.compare(g2), g2.compare(g), g.count_common(g2), g.intersection_size(g2))
            exp = (o.compare(o), o.compare(o2), o2.compare(o), o.count_common(o2), o.intersection_size(o2))
            if got != exp and not (all(x != x for x in (got[0], exp[0]))):
                print("PAIRWISE MISMATCH case", case, "step", step, got, exp)
                sys.exit(1)
            eg, eo = run_both(g, o, lambda m: m.merge(g2 if m is g else o2))
            if eg != eo or state(g) != state(o):
                print("MERGE MISMATCH case", case, eg, eo)
                sys.exit(1)
        n_sk += 1
    elif rng.random() < 0.12:
        # the SHARDED all-vs-all matrix: `world` ranks played one after the other (distributed.simulate_sharded: the sliced
        # dictionary, pair ownership, the mirrored-block exchange) on a ragged collection, whole matrix against the oracle
        from sourmash_rust_amd import distributed as D
        nrs = np.random.RandomState(rng.getrandbits(31))
        n = rng.randint(2, 260)
        world = rng.choice([2, 3, 4, 5, 8])
        width = rng.choice([8, 60, 300])
        shift = rng.choice([0, 0, 20, 40])                     # hash values crowded into a corner of hash space
        pool = np.unique(nrs.randint(0, 1 << 62, size=width * rng.choice([2, 6, 30]), dtype=np.int64).astype(np.uint64) >> np.uint64(shift))
        sks = [np.sort(nrs.choice(pool, min(len(pool), int(nrs.choice([0, 1, width // 2, width, 2 * width]))), replace=False)) for _ in range(n)]
        if rng.random() < 0.3:
            sks = [np.unique(np.concatenate([x, np.array([7, 99], dtype=np.uint64)])) if len(x) and nrs.rand() < 0.8 else x for x in sks]
        num = rng.choice([0, width, width // 2, 3])
        flat, off = pkg.matrix.csr_from_sketches(sks)
        if flat.size == 0:
            continue
        t = torch.from_numpy(flat.view(np.int64)).cuda()
        tune = rng.choice([dict(), dict(), dict(route="components"), dict(route="tiled"), dict(split_frequent=False)])
        want = ("jaccard", "common", "size") + (("count_common", "containment") if rng.random() < 0.5 else ())
        with pkg.matrix.tuning(**tune):
            outs = D.simulate_sharded((t, off), n, num, world, want=want)
        common, size, jac = coracle.compare_matrix(sks, sks, num, 31, 0 if num else 1 << 62)
        got = {k: torch.cat([o[k] for o in outs]).cpu().numpy() for k in want}
        ok = (got["common"].view(np.uint64) == common).all() and (got["size"].view(np.uint64) == size).all() and (got["jaccard"] == jac).all()
        if ok and "count_common" in want:
            cc = np.array([[len(np.intersect1d(a, b)) for b in sks] for a in sks], dtype=np.int64).reshape(n, n)
            lens = np.array([len(x) for x in sks], dtype=np.float64)[:, None]
            with np.errstate(invalid="ignore", divide="ignore"):
                cont = cc / lens
            ok = (got["count_common"] == cc).all() and ((got["containment"] == cont) | (np.isnan(cont) & np.isnan(got["containment"]))).all()
        if not ok:
            print("SHARDED MATRIX MISMATCH", n, world, width, shift, num, tune, want)
            sys.exit(1)
        n_cmp += 1
    elif rng.random() < 0.15:
        # a block with more than 96 Ki sharing pairs: the shape-based default takes the tiled kernel.
        # Device CSR in, whole matrix against the C oracle's compare_matrix.
        nrs = np.random.RandomState(rng.getrandbits(31))
        nrow, ncol = rng.randint(520, 700), rng.randint(520, 900)
        width = rng.choice([40, 200, 600])
        pool = np.unique(nrs.randint(0, 1 << 62, size=width * rng.choice([2, 5, 20]), dtype=np.int64).astype(np.uint64))
        ragged = rng.random() < 0.5
        def mkb(cnt):
            return [np.sort(nrs.choice(pool, min(len(pool), width if not ragged else nrs.choice([1, width // 3, width, 2 * width])),
                                       replace=False)) for _ in range(cnt)]
        rows = mkb(nrow)
        same = rng.random() < 0.4
        cols = rows if same else mkb(ncol)
        num = rng.choice([0, width, width // 2, 7])
        rf, ro = pkg.matrix.csr_from_sketches(rows)
        rt = torch.from_numpy(rf.view(np.int64)).cuda()
        if same:
            ct, co = rt, ro
        else:
            cf, co = pkg.matrix.csr_from_sketches(cols)
            ct = torch.from_numpy(cf.view(np.int64)).cuda()
        tune = rng.choice([dict(), dict(), dict(route="tiled", visit_all_tiles=True), dict(route="tiled", use_symmetry=False)])
        with pkg.matrix.tuning(**tune):
            out = pkg.matrix.compare_block_dev(rt, ro, ct, co, num, want=("jaccard", "common", "size"))
            st = pkg.matrix.last_stats()
        common, size, jac = coracle.compare_matrix(rows, cols, num, 31, 0 if num else 1 << 62)
        if st["route"] != "tiled" or not ((out["common"].cpu().numpy().view(np.uint64) == common).all()
                                          and (out["size"].cpu().numpy().view(np.uint64) == size).all()
                                          and (out["jaccard"].cpu().numpy() == jac).all()):
            print("BIG COMPARE MISMATCH", nrow, ncol, width, num, same, ragged, tune, st)
            sys.exit(1)
        n_cmp += 1
    else:
        uni = np.unique(np.array([rng.getrandbits(63) for _ in range(rng.choice([50, 2000, 20000]))], dtype=np.uint64))
        nrow, ncol = rng.randint(1, 90), rng.randint(1, 140)
        if rng.random() < 0.3:      # the few-vs-many shape (find / scaffold), either orientation
            nrow, ncol = rng.randint(1, 6), rng.randint(64, 300)
            if rng.random() < 0.5:
                nrow, ncol = ncol, nrow
        def mk(cnt):
            out = []
            for _ in range(cnt):
                sz = min(len(uni), rng.choice([0, 1, 3, 50, 300, 300, 1500]))
                out.append(np.sort(np.random.RandomState(rng.getrandbits(31)).choice(uni, sz, replace=False)))
            return out
        rows, cols = mk(nrow), mk(ncol)
        if rng.random() < 0.4 and nrow + ncol >= 32:      # frequent hashes: held by most sketches (set aside by the block compare)
            freq = np.unique(np.array([rng.getrandbits(63) for _ in range(rng.choice([1, 2, 5, 64, 80]))], dtype=np.uint64))
            def add(lst):
                return [np.unique(np.concatenate([x, freq[np.random.RandomState(rng.getrandbits(31)).random_sample(len(freq)) < 0.8]]))
                        if len(x) else x for x in lst]
            rows, cols = add(rows), add(cols)
        nums = [rng.choice([0, 1, 10, 300, 5000]) for _ in rows]
        same_num = rng.random() < 0.5
        if same_num:
            nums = [nums[0]] * nrow
        gm, om = [], []
        for r, nn in zip(rows, nums):
            a, b = pkg.KmerMinHash(nn, 21, False, 42, 0), coracle.MinHash(nn, 21, False, 42, 0)
            for h in r:
                a.mins_push(int(h)); b.mins_push(int(h))
            gm.append(a); om.append(b)
        gc, oc = [], []
        for c in cols:
            a, b = pkg.KmerMinHash(7, 21, False, 42, 0), coracle.MinHash(7, 21, False, 42, 0)
            for h in c:
                a.mins_push(int(h)); b.mins_push(int(h))
            gc.append(a); oc.append(b)
        if rng.random() < 0.25:        # all-vs-all of one list (symmetric when the nums agree)
            gc, oc, cols, ncol = gm, om, rows, nrow
        want_cc = rng.random() < 0.6   # without count_common the kernels take their early-exit instantiation
        # any way the block can be served must give the same numbers (smh_compare_set_tuning)
        tune = rng.choice([dict(), dict(), dict(route="components"), dict(route="tiled"), dict(route="tiled", visit_all_tiles=True),
                           dict(route="tiled", visit_all_tiles=True, use_symmetry=False), dict(comp_pairs_limit=0),
                           dict(split_frequent=False), dict(route="tiled", split_frequent=False)])
        with pkg.matrix.tuning(**tune):
            out = pkg.matrix.compare_block(gm, gc, want=("jaccard", "common", "size") + (("count_common",) if want_cc else ()))
        for i in range(nrow):
            for j in range(ncol):
                c, s_ = om[i].intersection_size(oc[j])
                if (int(out["common"][i, j]), int(out["size"][i, j])) != (c, s_) or out["jaccard"][i, j] != om[i].compare(oc[j]) \
                        or (want_cc and int(out["count_common"][i, j]) != om[i].count_common(oc[j])):
                    print("COMPARE MISMATCH", nrow, ncol, i, j, nums[i], len(rows[i]), len(cols[j]))
                    sys.exit(1)
        n_cmp += 1
print("fuzz ok: %d sketch cases, %d compare blocks, seed %d" % (n_sk, n_cmp, seed))
