"""Randomised parity of the tiled route on MID-SIZE and LARGE blocks (16- and 32-row tiles of k_compare_tiled_pf), which the
fuzz's small blocks do not reach: ragged sketches, a few families, random num (0 = no cut), same set and two sets, every
output.  Sampled rows x all columns against the C oracle; symmetry when rows == columns.
    python tests/stress_tiled.py <seconds> <seed>      (run by tests/test_gpu_compare.py for 20 seconds; the C oracle is the checker)"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
from __graft_entry__ import load_package
pkg = load_package()
import coracle  # noqa: E402  (the checker: tools/ and tests/ only)
coracle.build()
from sourmash_rust_amd import matrix as MX

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
t_end = time.time() + budget
done = 0
shapes = {}
while time.time() < t_end:
    nrs = np.random.RandomState(rng.getrandbits(31))
    n = rng.choice([1300, 1700, 2200, 2800, 3600])
    fams = rng.choice([1, 1, 2, 5])
    base_len = rng.choice([300, 1000, 2000])
    num = rng.choice([0, base_len // 2, base_len, 3 * base_len])
    pools = [np.unique(nrs.randint(0, 1 << 62, size=int(base_len * rng.choice([1.5, 3, 8])), dtype=np.int64).astype(np.uint64)) for _ in range(fams)]
    sks = []
    for i in range(n):
        L = int(base_len * rng.choice([0.25, 1, 1, 1, 2])) if rng.random() < 0.3 else base_len
        pool = pools[i % fams]
        k = min(L, pool.size)
        own = nrs.randint(0, 1 << 62, size=max(1, L // 5), dtype=np.int64).astype(np.uint64)
        s = np.unique(np.concatenate([nrs.choice(pool, k - k // 5, replace=False), own]))
        if num:
            s = s[:max(1, min(s.size, num if rng.random() < 0.8 else s.size))]
        sks.append(s)
    two_sets = rng.random() < 0.25
    rows = sks if not two_sets else sks[: n // 2]
    cols = sks if not two_sets else sks[n // 3:]
    rflat, roff = MX.csr_from_sketches(rows)
    rt = torch.from_numpy(rflat.view(np.int64)).cuda()
    if two_sets:
        cflat, coff = MX.csr_from_sketches(cols)
        ct = torch.from_numpy(cflat.view(np.int64)).cuda()
    else:
        ct, coff = rt, roff
    want = ("jaccard", "common", "size", "count_common", "containment")
    tune = rng.choice([dict(route="tiled"), dict(route="tiled"), dict(), dict(route="tiled", use_symmetry=False)])
    with MX.tuning(**tune):
        out = MX.compare_block_dev(rt, roff, ct, coff, num, want=want)
    st = MX.last_stats()
    shapes[(st["route"], st["rows_per_tile"])] = shapes.get((st["route"], st["rows_per_tile"]), 0) + 1
    pick = sorted(set([0, len(rows) - 1] + [rng.randrange(len(rows)) for _ in range(10)]))
    common, size, jac = coracle.compare_matrix([rows[i] for i in pick], cols, num, 31, 0)
    idx = torch.tensor(pick, device="cuda")
    ok = (out["common"][idx].cpu().numpy().view(np.uint64) == common).all() and (out["size"][idx].cpu().numpy().view(np.uint64) == size).all()
    jg = out["jaccard"][idx].cpu().numpy()
    ok = ok and ((jg == jac) | (np.isnan(jg) & np.isnan(jac))).all()
    cc = out["count_common"][idx].cpu().numpy().view(np.uint64)
    for a, i in enumerate(pick[:3]):
        for j in range(0, len(cols), max(1, len(cols) // 40)):
            ok = ok and int(cc[a, j]) == len(np.intersect1d(rows[i], cols[j]))
    if not two_sets:
        j = out["jaccard"]
        ok = ok and bool(((j == j.T) | (torch.isnan(j) & torch.isnan(j.T))).all())
    if not ok:
        print("MISMATCH", dict(n=n, fams=fams, base_len=base_len, num=num, two_sets=two_sets, tune=tune), st)
        sys.exit(1)
    done += 1
    del out
print("stress ok: %d blocks, seed %d, (route, rows per tile) seen: %s" % (done, seed, sorted(shapes.items())))
