"""Deterministic synthetic inputs of SURVEY.md 8(d): family-structured MinHash signatures for the
compare matrix (C3/C4).  Counter-based (splitmix64), so any slice of the global signature list can
be generated independently on any rank."""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, idx):
    """Vectorised splitmix64 output for counters idx (uint64 array) under `seed`."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def family_signatures(lo, hi, num=2000, n_families=50, pool=4000, keep=0.8, private=1000, seed=3):
    """Signatures lo..hi-1 of the global list: signature i belongs to family i % n_families and is the
    bottom-`num` of (each of the family's `pool` hashes kept w.p. `keep`) u (`private` own hashes).
    Returns an (hi-lo, num) uint64 array, rows ascending and distinct."""
    out = np.empty((hi - lo, num), dtype=np.uint64)
    j_pool = np.arange(pool, dtype=np.uint64)
    j_priv = np.arange(private, dtype=np.uint64)
    thresh = np.uint64(int(keep * 2.0 ** 64) - 1) if keep < 1.0 else M64
    for r, i in enumerate(range(lo, hi)):
        fam = i % n_families
        ph = splitmix64(seed * 1000003 + fam * 7919 + 1, j_pool)
        mask = splitmix64(seed * 1000003 + 500000 + i * 2 + 0, j_pool) <= thresh
        pv = splitmix64(seed * 1000003 + 500000 + i * 2 + 1, j_priv)
        u = np.unique(np.concatenate([ph[mask], pv]))
        if u.size < num:
            raise ValueError("signature %d has only %d hashes" % (i, u.size))
        out[r] = u[:num]
    return out
