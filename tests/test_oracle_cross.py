"""The two independent restatements (C and Python) must agree on random inputs, including the
arms no reference test pins (protein, abundance quirk Q3, num+max_hash quirk Q4, merge quirk Q5)."""
import random

import pytest

ALPH = b"ACGTacgtNnRXY*-"


def rand_seq(rng, n, bad=0.02):
    out = bytearray()
    for _ in range(n):
        if rng.random() < bad:
            out.append(rng.choice(ALPH))
        else:
            out.append(rng.choice(b"ACGT" if rng.random() < 0.9 else b"acgt"))
    return bytes(out)


def same_state(c, p):
    assert c.mins == p.mins
    assert c.abunds == p.abunds


CASES = [
    # num, ksize, prot, seed, max_hash, track
    (20, 10, False, 42, 0, False),
    (20, 10, False, 42, 0, True),
    (5, 4, False, 42, 0, True),
    (0, 7, False, 42, 1 << 61, True),
    (0, 21, False, 42, 1 << 58, False),
    (8, 5, False, 7, 1 << 62, True),          # Q4: num and max_hash both set
    (50, 9, True, 42, 0, True),
    (0, 27, True, 42, 1 << 60, True),
    (30, 12, True, (1 << 40) + 5, 0, False),   # seed >= 2^32 (Q10: u64 reading)
    (500, 31, False, 42, 0, False),
]


@pytest.mark.parametrize("case", CASES)
def test_add_sequence_agree(case, coracle, pyoracle):
    rng = random.Random(hash(case) & 0xFFFF)
    for trial in range(6):
        c = coracle.MinHash(*case)
        p = pyoracle.MinHash(*case)
        for _ in range(rng.randint(1, 3)):
            n = rng.choice([0, 1, case[1] - 1, case[1], case[1] + 1, 40, 200, 700])
            seq = rand_seq(rng, max(0, n), bad=rng.choice([0.0, 0.0, 0.02]))
            force = rng.random() < 0.5
            ec = ep = None
            try:
                c.add_sequence(seq, force)
            except Exception as e:
                ec = (getattr(e, "code", None), getattr(e, "message", ""))
            try:
                p.add_sequence(seq, force)
            except pyoracle.OracleError as e:
                ep = (e.code, e.message.split(": ")[-1])
            assert ec == ep
            same_state(c, p)


def test_small_alphabet_collisions(coracle, pyoracle):
    # few distinct k-mers -> many repeats: exercises the abundance paths hard
    rng = random.Random(5)
    for num, mx in [(3, 0), (6, 0), (0, 1 << 63), (4, 1 << 63)]:
        c = coracle.MinHash(num, 3, False, 42, mx, True)
        p = pyoracle.MinHash(num, 3, False, 42, mx, True)
        for _ in range(20):
            seq = bytes(rng.choice(b"ACGT") for _ in range(rng.randint(3, 30)))
            c.add_sequence(seq, True)
            p.add_sequence(seq, True)
            same_state(c, p)


def test_merge_compare_agree(coracle, pyoracle):
    rng = random.Random(11)
    for trial in range(40):
        num = rng.choice([0, 5, 20, 50])
        mx = 0 if num else 1 << 62
        ta, tb = rng.random() < 0.5, rng.random() < 0.5
        universe = [rng.getrandbits(62) for _ in range(60)]
        ca, pa = coracle.MinHash(num, 21, False, 42, mx, ta), pyoracle.MinHash(num, 21, False, 42, mx, ta)
        cb, pb = coracle.MinHash(num + rng.choice([0, 0, 3]), 21, False, 42, mx, tb), None
        pb = pyoracle.MinHash(cb.num, 21, False, 42, mx, tb)
        for h in rng.choices(universe, k=rng.randint(0, 80)):
            ca.add_hash(h); pa.add_hash(h)
        for h in rng.choices(universe, k=rng.randint(0, 80)):
            cb.add_hash(h); pb.add_hash(h)
        assert ca.count_common(cb) == pa.count_common(pb)
        assert ca.intersection_size(cb) == pa.intersection_size(pb)
        assert ca.compare(cb) == pa.compare(pb)
        assert cb.compare(ca) == pb.compare(pa)
        if pa.mins:
            assert ca.containment(cb) == pa.containment(pb)
        ca.merge(cb); pa.merge(pb)
        same_state(ca, pa)


def test_incompatible(coracle, pyoracle):
    for O in (coracle, pyoracle):
        base = O.MinHash(10, 21, False, 42, 0)
        for other, code in [(O.MinHash(10, 31, False, 42, 0), 101), (O.MinHash(10, 21, True, 42, 0), 102),
                            (O.MinHash(10, 21, False, 42, 5), 103), (O.MinHash(10, 21, False, 43, 0), 104)]:
            for fn in ("compare", "count_common", "merge", "intersection_size"):
                with pytest.raises(Exception) as ei:
                    getattr(base, fn)(other)
                assert ei.value.code == code


def test_synth_dna_agree(coracle, pyoracle):
    for start, n, seed, ne in [(0, 100, 1, 0), (31, 200, 2, 0), (99990, 64, 2, 100000), (5, 70, 9, 7)]:
        assert bytes(coracle.synth_dna(start, n, seed, ne)) == pyoracle.synth_dna(start, n, seed, ne)


def test_translate_frames(coracle, pyoracle):
    import ctypes as C
    rng = random.Random(3)
    L = coracle.lib()
    for _ in range(20):
        seq = rand_seq(rng, rng.randint(0, 60), bad=0.1)
        up = seq.upper() if all(c < 128 for c in seq) else seq
        for frame in range(3):
            for rc in (0, 1):
                buf = C.create_string_buffer(len(seq) + 4)
                n = C.c_size_t()
                L.omh_translate_frames(seq, len(seq), frame, rc, buf, C.byref(n))
                src = pyoracle.revcomp(up) if rc else up
                assert buf.raw[:n.value] == pyoracle.to_aa(src[frame:])
