"""GPU parity: KmerMinHash::add_sequence through the C ABI vs the oracle, bit-exact.

Covers the reference's own KATs, the committed golden vectors, and seeded random inputs over the
edge cases the reference handles: lowercase, invalid bytes with force on/off (partial state +
error k-mer), length < k / == k, every sketch mode (num, scaled, both, neither), abundance quirk
Q3, multi-call accumulation, k <= 32 (rolling kernel) and k > 32 (byte-wise kernel), the protein
arm with dropped codons, many records per launch, and device-resident input."""
import gzip
import json
import os
import random

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

MERGE_MINS = [
    2996412506971915891, 4448613756639084635, 8373222269469409550, 9390240264282449587,
    11085758717695534616, 11668188995231815419, 11760449009842383350, 14682565545778736889,
]


def same_state(g, o):
    assert g.mins == o.mins
    assert g.abunds == o.abunds


def test_reference_kats(pkg):
    assert pkg.hash_murmur(b"ACG", 42) == 1731421407650554201          # tests/test.rs:5
    mh = pkg.KmerMinHash(1, 4)
    with pytest.raises(pkg.SourmashError) as ei:                          # tests/minhash.rs:5-17
        mh.add_sequence(b"ATGR", False)
    assert ei.value.code == 1101 and ei.value.message == "invalid DNA character in input k-mer: ATGR"
    a, b = pkg.KmerMinHash(20, 10), pkg.KmerMinHash(20, 10)              # tests/minhash.rs:19-52
    a.add_sequence(b"TGCCGCCCAGCA"); b.add_sequence(b"TGCCGCCCAGCA")
    a.add_sequence(b"GTCCGCCCAGTGA"); b.add_sequence(b"GTCCGCCCAGTGG")
    a.merge(b)
    assert a.mins == MERGE_MINS
    s1 = b"TGCCGCCCAGCACCGGGTGACTAGGTTGAGCCATGATTAACCTGCAATGA"           # tests/minhash.rs:54-83
    s2 = b"GATTGGTGCACACTTAACTGGGTGCCGCGCTGGTGCTGATCCATGAAGTT"
    a, b = pkg.KmerMinHash(20, 10), pkg.KmerMinHash(20, 10)
    a.add_sequence(s1); b.add_sequence(s1)
    assert a.compare(b) == 1.0 and b.compare(a) == 1.0
    b.add_sequence(s1)
    assert a.compare(b) == 1.0 and b.compare(a) == 1.0
    b.add_sequence(s2)
    assert a.compare(b) >= 0.3 and b.compare(a) >= 0.3


def test_hash_words(pkg, coracle):
    rng = random.Random(1)
    words = [bytes(rng.getrandbits(8) | 1 for _ in range(n)) for n in list(range(0, 40)) + [63, 64, 65, 200]]
    got = pkg.hash_words(words, 42)
    assert [int(x) for x in got] == [coracle.hash_murmur(w, 42) for w in words]
    big_seed = (1 << 40) + 12345
    assert int(pkg.hash_words([b"ACGTACGT"], big_seed)[0]) == coracle.hash_murmur(b"ACGTACGT", big_seed)
    mh, o = pkg.KmerMinHash(5, 3), coracle.MinHash(5, 3)
    for w in (b"AAA", b"ACG", b"TTT", b"whatever"):
        mh.add_word(w); o.add_word(w)
    same_state(mh, o)


def golden():
    with gzip.open(os.path.join(GOLDEN, "golden_vectors.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", sorted(golden()["sketch"]))
def test_golden_vectors(name, pkg, coracle):
    g = golden()["sketch"][name]
    mh = pkg.KmerMinHash(*g["params"])
    mh.add_sequence(bytes(coracle.synth_dna(*g["synth"])), g["force"])
    assert mh.mins == g["mins"] and mh.abunds == g["abunds"]


ALPH = b"ACGTacgtNnRXY*-"


def rand_seq(rng, n, bad=0.02):
    out = bytearray()
    for _ in range(n):
        if rng.random() < bad:
            out.append(rng.choice(ALPH))
        else:
            out.append(rng.choice(b"ACGT" if rng.random() < 0.9 else b"acgt"))
    return bytes(out)


CASES = [
    # num, ksize, prot, seed, max_hash, track
    (20, 10, False, 42, 0, False),
    (20, 10, False, 42, 0, True),
    (5, 4, False, 42, 0, True),
    (0, 7, False, 42, 1 << 61, True),
    (0, 21, False, 42, 1 << 58, False),
    (8, 5, False, 7, 1 << 62, True),           # Q4: num and max_hash both set (order-dependent)
    (0, 6, False, 42, 0, True),                # neither set (order-dependent)
    (500, 31, False, 42, 0, False),
    (100, 31, False, 42, 0, True),
    (50, 32, False, 42, 0, True),
    (50, 1, False, 42, 0, True),
    (64, 16, False, 42, 0, True),
    (64, 17, False, 9, 0, False),
    (30, 33, False, 42, 0, True),              # byte-wise kernel
    (30, 51, False, 42, 0, True),
    (0, 51, False, 42, 1 << 60, True),
    (20, 48, False, 42, 0, True),              # 4-limb rolling kernel, run-time k
    (20, 63, False, 42, 0, False),
    (20, 64, False, 42, 0, True),
    (20, 65, False, 42, 0, True),              # 8-limb rolling kernel
    (10, 100, False, 42, 0, False),
    (10, 128, False, 42, 0, True),
    (10, 129, False, 42, 0, True),             # byte-wise kernel (k > 128)
    (5, 300, False, 42, 0, False),
    (50, 9, True, 42, 0, True),
    (0, 27, True, 42, 1 << 60, True),
    (30, 12, True, (1 << 40) + 5, 0, False),
    (40, 21, False, (1 << 40) + 5, 0, False),  # seed >= 2^32 (Q10)
]


@pytest.mark.parametrize("case", CASES)
def test_add_sequence_random(case, pkg, coracle):
    rng = random.Random(hash(case) & 0xFFFF)
    k = case[1]
    for trial in range(5):
        g = pkg.KmerMinHash(*case)
        o = coracle.MinHash(*case)
        for _ in range(rng.randint(1, 3)):
            n = rng.choice([0, 1, k - 1, k, k + 1, 40, 200, 3000, 20000])
            seq = rand_seq(rng, max(0, n), bad=rng.choice([0.0, 0.0, 0.01]))
            force = rng.random() < 0.5
            eg = eo = None
            try:
                g.add_sequence(seq, force)
            except pkg.SourmashError as e:
                eg = (e.code, e.message.split(": ")[-1] if e.code == 1101 else "")
            try:
                o.add_sequence(seq, force)
            except coracle.OracleError as e:
                eo = (e.code, e.message if e.code == 1101 else "")
            assert eg == eo
            same_state(g, o)


def test_few_distinct_kmers_abundance(pkg, coracle):
    # heavy repetition: the abundance closed form (Q3) against one-by-one insertion
    rng = random.Random(5)
    for num, mx in [(3, 0), (6, 0), (40, 0), (0, 1 << 63), (4, 1 << 63)]:
        g = pkg.KmerMinHash(num, 3, False, 42, mx, True)
        o = coracle.MinHash(num, 3, False, 42, mx, True)
        for _ in range(12):
            seq = bytes(rng.choice(b"ACGT") for _ in range(rng.choice([3, 10, 30, 500, 70000])))
            g.add_sequence(seq, True); o.add_sequence(seq, True)
            same_state(g, o)


def test_num_mode_multi_chunk_abundance(pkg, coracle):
    # longer than the first num-mode chunk, so thresholds and T* cross chunk boundaries
    for num, k, track in [(50, 12, True), (500, 31, True), (2000, 21, False)]:
        seq = bytes(coracle.synth_dna(0, 400000, 11 + num, 0))
        seq = seq[:150000] + seq[:100000] + seq[150000:]   # repeats across chunks
        g = pkg.KmerMinHash(num, k, False, 42, 0, track)
        o = coracle.MinHash(num, k, False, 42, 0, track)
        g.add_sequence(seq, True); o.add_sequence(seq, True)
        same_state(g, o)
        g.add_sequence(seq[1000:90000], True); o.add_sequence(seq[1000:90000], True)
        same_state(g, o)


def test_c1_config_sketch_and_self_compare(pkg, coracle):
    # BASELINE config 0: 1 MB DNA, k=31, num=500, compare to itself -> 1.0
    seq = bytes(coracle.synth_dna(0, 1000000, 1, 0))
    g = pkg.KmerMinHash(500, 31)
    o = coracle.MinHash(500, 31)
    g.add_sequence(seq); o.add_sequence(seq)
    same_state(g, o)
    assert g.compare(g) == 1.0 and g.count_common(g) == 500


def test_nul_bytes_and_high_bytes(pkg, coracle):
    seq = b"ACGTACGTAC\x00ACGTTTGACA\xffGGGATCCAT\xc3\xa9ACGATCGATTTTACG"
    for force in (True, False):
        for case in [(10, 5, False, 42, 0, True), (10, 6, True, 42, 0, True)]:
            g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
            eg = eo = None
            try:
                g.add_sequence(seq, force)
            except pkg.SourmashError as e:
                eg = e.code
            try:
                o.add_sequence(seq, force)
            except coracle.OracleError as e:
                eo = e.code
            assert eg == eo
            same_state(g, o)


def test_protein_dropped_codons(pkg, coracle):
    rng = random.Random(8)
    for trial in range(6):
        seq = bytearray(rand_seq(rng, rng.choice([30, 100, 1000, 30000]), bad=0.0))
        for _ in range(rng.randint(0, 6)):     # runs of N splice residues together (Q8)
            p = rng.randrange(len(seq))
            seq[p:p + rng.randint(1, 12)] = b"N" * rng.randint(1, 12)
        for case in [(40, 21, True, 42, 0, True), (0, 30, True, 42, 1 << 61, True), (30, 3, True, 42, 0, False)]:
            g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
            g.add_sequence(bytes(seq), True); o.add_sequence(bytes(seq), True)
            same_state(g, o)


def test_protein_arm_pinned_to_reference_codontable(pkg, coracle):
    """The device's translation against DATA of the reference: tests/golden/codontable.json holds the
    64 pairs of the CODONTABLE literal (reference src/lib.rs:691-777) and the six-frame order of
    src/lib.rs:280-300 (tests/golden/make_codontable.py).  Expected sketches are built here from that
    table alone; only the hash of a residue window and add_hash come from the oracle, and both of
    those are pinned by reference KATs (tests/test_oracle_kat.py)."""
    from test_codontable import revcomp, table_windows
    with open(os.path.join(GOLDEN, "codontable.json")) as fh:
        doc = json.load(fh)
    # (1) every codon on its own: "XYZ" -> forward frame 0 gives table[XYZ], reverse frame 0 gives
    # table[revcomp(XYZ)], frames 1 and 2 are incomplete.  ksize 3 = windows of one residue.
    for codon, aa in sorted(doc["table"].items()):
        g = pkg.KmerMinHash(0, 3, True, 42, (1 << 64) - 1, True)
        g.add_sequence(codon.encode(), True)
        exp = {}
        for r in (aa, doc["table"][revcomp(codon.encode()).decode()]):
            h = coracle.hash_murmur(r.encode(), 42)
            exp[h] = exp.get(h, 0) + 1
        assert dict(zip(g.mins, g.abunds)) == exp, codon
    # (2) all 64 codons in one record, all six frames, every ksize/3 in 1..4, lower case included
    allc = "".join(sorted(doc["table"])).encode()
    rng = random.Random(77)
    for seq in (allc, allc.lower(), bytes(rng.choice(b"ACGT") for _ in range(5000)),
                bytes(rng.choice(b"ACGTN") for _ in range(3001))):
        for ksize in (3, 6, 9, 12, 27):
            for num, mx in ((0, (1 << 64) - 1), (25, 0)):      # scaled: total counts; num + abundance: order matters (Q3)
                g = pkg.KmerMinHash(num, ksize, True, 42, mx, True)
                o = coracle.MinHash(num, ksize, True, 42, mx, True)
                g.add_sequence(seq, True)
                for w in table_windows(doc, seq, ksize):
                    o.add_word(w)
                same_state(g, o)


def _profile(pkg, name):
    import ctypes as C
    ms, n = C.c_double(), C.c_uint64()
    pkg.lib().smh_profile_get(name.encode(), C.byref(ms), C.byref(n))
    return n.value


@pytest.mark.parametrize("ksize", [21, 27, 28, 29, 30, 32])
def test_protein_fused_kernel(ksize, pkg, coracle):
    """The one-pass protein kernel (translation + hashing, no residue buffer; window lengths 7, 9, 10)
    against the oracle on what it has to get right by itself: many records per launch with lengths
    around ksize and around the tile geometry, lower case, runs of N that splice residues together
    (quirk Q8), N at record ends, every sketch mode with abundance (positions = the reference's frame
    order, quirk Q3), and bytes >= 0x80, for which the launch is discarded and repeated on the
    two-pass path."""
    rng = random.Random(ksize)
    L = pkg.lib()

    def rec(n, n_frac=0.0, lower=0.0):
        out = bytearray(rng.choice(b"ACGT") for _ in range(n))
        for i in range(n):
            if rng.random() < lower:
                out[i] |= 0x20
        k = int(n * n_frac)
        for _ in range(k):
            p = rng.randrange(max(1, n)); ln = rng.choice([1, 1, 2, 3, 7, 30])
            out[p:p + ln] = b"N" * min(ln, n - p)
        return bytes(out)

    lens = [0, 1, 2, ksize - 1, ksize, ksize + 1, ksize + 2, 3 * (ksize // 3) + 3, 100, 127, 128, 129, 1000, 4093, 70000, 65536 + 13]
    batches = [
        [rec(n) for n in lens],
        [rec(n, n_frac=0.01, lower=0.3) for n in lens + [30000]],
        [b"N" * 5 + rec(300) + b"NN", rec(64) + b"N", b"N" + rec(64), rec(29) + b"N" + rec(29) + b"NNN" + rec(31)],
        [rec(200000, n_frac=0.0005)],
    ]
    for case in [(0, ksize, True, 42, (1 << 64) // 50, True), (60, ksize, True, 42, 0, True), (0, ksize, True, 7, (1 << 64) - 1, False)]:
        for recs in batches:
            if case[4] == (1 << 64) - 1 and sum(map(len, recs)) > 40000:
                continue
            g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
            L.smh_profile_reset(); L.smh_profile_enable(1)
            g.add_sequences(recs, True)
            fused = _profile(pkg, "protein_fused")
            L.smh_profile_enable(0)
            for r in recs:
                o.add_sequence(r, True)
            same_state(g, o)
            if ksize // 3 in (7, 9, 10) and any(len(r) >= ksize for r in recs):
                assert fused >= 1, "the fused kernel did not run"
    # a non-ASCII byte: valid UTF-8 that is no codon is dropped, invalid UTF-8 panics (code 1) with the
    # frames before it kept -- the fused launch must notice and hand over
    for tail in (b"\xc3\xa9", b"\xff"):
        seq = rec(500) + tail + rec(500)
        g, o = pkg.KmerMinHash(0, ksize, True, 42, (1 << 64) // 20, True), coracle.MinHash(0, ksize, True, 42, (1 << 64) // 20, True)
        eg = eo = None
        try:
            g.add_sequence(seq, True)
        except pkg.SourmashError as e:
            eg = e.code
        try:
            o.add_sequence(seq, True)
        except coracle.OracleError as e:
            eo = e.code
        assert eg == eo
        same_state(g, o)


def test_many_records_per_launch(pkg, coracle):
    rng = random.Random(13)
    recs = [rand_seq(rng, rng.choice([0, 5, 30, 31, 32, 100, 151, 151, 151, 2000]), bad=rng.choice([0, 0, 0.01]))
            for _ in range(300)]
    for case in [(500, 31, False, 42, 0, True), (0, 21, False, 42, 1 << 59, True), (60, 15, True, 42, 0, True),
                 (0, 40, False, 42, 1 << 60, False)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        g.add_sequences(recs, True)
        for r in recs:
            o.add_sequence(r, True)
        same_state(g, o)
    # force=False: every record is processed up to its first bad window, the first error is reported
    g, o = pkg.KmerMinHash(100, 21, False, 42, 0, True), coracle.MinHash(100, 21, False, 42, 0, True)
    first = None
    for r in recs:
        try:
            o.add_sequence(r, False)
        except coracle.OracleError as e:
            first = first or e.message
    with pytest.raises(pkg.SourmashError) as ei:
        g.add_sequences(recs, False)
    assert ei.value.code == 1101 and ei.value.message.endswith(first)
    same_state(g, o)


def test_device_resident_input_and_generator(pkg, coracle):
    import ctypes as C
    import torch
    n = 3_000_000
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    pkg.lib().smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 64, n, 2, 100000, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    host = bytes(coracle.synth_dna(64, n, 2, 100000))
    assert bytes(buf.cpu().numpy()) == host
    offs = [0, 1_000_000, 1_000_000, 2_500_000, n]
    for case in [(0, 31, False, 42, 18446744073709552, True), (500, 31, False, 42, 0, False)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        g.add_sequences_dev(buf.data_ptr(), n, offs, True, torch.cuda.current_stream().cuda_stream)
        for a, b in zip(offs[:-1], offs[1:]):
            o.add_sequence(host[a:b], True)
        same_state(g, o)
    # an unaligned device pointer (slice) must work too
    g, o = pkg.KmerMinHash(0, 31, False, 42, 18446744073709552, True), coracle.MinHash(0, 31, False, 42, 18446744073709552, True)
    g.add_sequences_dev(buf.data_ptr() + 7, 500_001, [0, 500_001], True)
    o.add_sequence(host[7:500_008], True)
    same_state(g, o)


def test_full_size_properties(pkg, coracle):
    """Size-independent checks at a size the oracle cannot finish quickly (2 GB): chunking is
    exact (sketching two halves with k-1 overlap and merging == sketching the whole) and the
    abundance total equals the number of retained k-mer occurrences."""
    import ctypes as C
    import torch
    n = 2 * 1024 * 1024 * 1024
    mx = 18446744073709552
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    pkg.lib().smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 2, 0, C.c_void_p(0))
    torch.cuda.synchronize()
    whole = pkg.KmerMinHash(0, 31, False, 42, mx, True)
    whole.add_sequences_dev(buf.data_ptr(), n, [0, n], True)
    half = n // 2
    a = pkg.KmerMinHash(0, 31, False, 42, mx, True)
    b = pkg.KmerMinHash(0, 31, False, 42, mx, True)
    a.add_sequences_dev(buf.data_ptr(), half + 30, [0, half + 30], True)
    b.add_sequences_dev(buf.data_ptr() + half, n - half, [0, n - half], True)
    a.merge(b)
    assert (a.mins_np() == whole.mins_np()).all() and (a.abunds_np() == whole.abunds_np()).all()
    m = whole.mins_np()
    assert (m[1:] > m[:-1]).all() and m[-1] <= mx
    # expected retained fraction 1/1000 of n-30 windows (binomial, 6 sigma)
    tot = int(whole.abunds_np().sum())
    exp = (n - 30) / 1000.0
    assert abs(tot - exp) < 6 * exp ** 0.5
    # the first 2 MB agree with the oracle exactly
    o = coracle.MinHash(0, 31, False, 42, mx, True)
    o.add_sequence(bytes(coracle.synth_dna(0, 2_000_000, 2, 0)), True)
    g = pkg.KmerMinHash(0, 31, False, 42, mx, True)
    g.add_sequences_dev(buf.data_ptr(), 2_000_000, [0, 2_000_000], True)
    same_state(g, o)


@pytest.mark.parametrize("n_every", [0, 100000])
def test_c2_at_full_size(n_every, pkg, coracle):
    """BASELINE configs[1] at its full size: 10 GB as 10 000 records x 1 MB, k=31, scaled=1000,
    force=true -- clean, and with one `N` per 10^5 bases.  Size-independent
    properties: ascending distinct hashes <= max_hash; the retained total follows the binomial
    law of the number of valid windows; linearity (sketch(all) == merge(sketch(first 4 000
    records), sketch(rest)), abundances included); and sampled records -- the first, one in the
    middle, the last -- sketched by the C oracle must be subsets of the whole with abundances no
    larger than the whole's (equality for the hashes that occur in no other record is implied by
    linearity + the per-record parity tests)."""
    import ctypes as C
    import torch
    nrec, rlen, mx = 10000, 1_000_000, 18446744073709552
    total = nrec * rlen
    buf = torch.empty(total, dtype=torch.uint8, device="cuda")
    assert pkg.lib().smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 2, n_every, C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    off = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(rlen)

    def sketch(r0, r1):
        mh = pkg.KmerMinHash(0, 31, False, 42, mx, True)
        mh.add_sequences_dev(buf.data_ptr() + r0 * rlen, (r1 - r0) * rlen, off[r0:r1 + 1] - off[r0], True)
        return mh

    whole = sketch(0, nrec)
    m, ab = whole.mins_np(), whole.abunds_np()
    assert (m[1:] > m[:-1]).all() and m[-1] <= mx
    # valid windows per record: the generator puts an N at every position = n_every - 1 (mod n_every) and
    # n_every divides the record length, so a record is rlen / n_every runs of n_every - 1 valid bases
    windows = nrec * ((rlen // n_every) * (n_every - 1 - 30) if n_every else rlen - 30)
    exp = windows * ((mx + 1) / 2.0 ** 64)
    assert abs(int(ab.sum()) - exp) < 6 * exp ** 0.5
    a, b = sketch(0, 4000), sketch(4000, nrec)
    a.merge(b)
    assert (a.mins_np() == m).all() and (a.abunds_np() == ab).all()
    whole_map = dict(zip(m.tolist(), ab.tolist()))
    for r in (0, 4999, nrec - 1):
        o = coracle.MinHash(0, 31, False, 42, mx, True)
        o.add_sequence(bytes(coracle.synth_dna(r * rlen, rlen, 2, n_every)), True)
        g = sketch(r, r + 1)
        same_state(g, o)
        assert all(whole_map.get(h, 0) >= c for h, c in zip(o.mins, o.abunds))


def test_c5_share_at_full_size(pkg, coracle):
    """BASELINE configs[4], one rank's share: 12.5 GB of DNA (12 500 records x 1 MB) through the
    protein arm, ksize=27, scaled=1000, abundance tracking -- the one-pass kernel at full size.
    Properties: ascending, <= max_hash; binomial total over the 2 x 3 x 12 500 frame windows;
    linearity across a split; two sampled records equal to the C oracle's sketch and contained in the whole."""
    import ctypes as C
    import torch
    nrec, rlen, mx = 12500, 1_000_000, 18446744073709552
    total = nrec * rlen
    buf = torch.empty(total + 64, dtype=torch.uint8, device="cuda")
    assert pkg.lib().smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 5, 0, C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    off = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(rlen)

    def sketch(r0, r1):
        mh = pkg.KmerMinHash(0, 27, True, 42, mx, True)
        mh.add_sequences_dev(buf.data_ptr() + r0 * rlen, (r1 - r0) * rlen, off[r0:r1 + 1] - off[r0], True)
        return mh

    L = pkg.lib()
    L.smh_profile_reset(); L.smh_profile_enable(1)
    whole = sketch(0, nrec)
    fused = _profile(pkg, "protein_fused")
    L.smh_profile_enable(0)
    assert fused == 1 and _profile(pkg, "translate") == 0, "the share must take the one-pass kernel"
    m, ab = whole.mins_np(), whole.abunds_np()
    assert (m[1:] > m[:-1]).all() and m[-1] <= mx
    windows = sum(2 * ((rlen - f) // 3 - 9 + 1) for f in range(3)) * nrec
    exp = windows * ((mx + 1) / 2.0 ** 64)
    assert abs(int(ab.sum()) - exp) < 6 * exp ** 0.5
    a, b = sketch(0, 6000), sketch(6000, nrec)
    a.merge(b)
    assert (a.mins_np() == m).all() and (a.abunds_np() == ab).all()
    whole_map = dict(zip(m.tolist(), ab.tolist()))
    for r in (0, nrec - 1):
        o = coracle.MinHash(0, 27, True, 42, mx, True)
        o.add_sequence(bytes(coracle.synth_dna(r * rlen, rlen, 5, 0)), True)
        g = sketch(r, r + 1)
        same_state(g, o)
        assert all(whole_map.get(h, 0) >= c for h, c in zip(o.mins, o.abunds))


def test_candidate_buffer_overflow_is_rerun(pkg, coracle):
    """Far more survivors than the uniform-hash estimate: poly-A where the single k-mer passes the
    filter.  The first launch overflows its buffer (the counter keeps counting), the library re-runs
    it with the exact size, and the LDS stage spills to the global sink."""
    k = 21
    h = int(pkg.hash_words([b"A" * k], 42)[0])
    n = 3_000_000
    seq = b"A" * n
    for num, mx in [(0, h), (0, h - 1), (10, 0)]:
        g = pkg.KmerMinHash(num, k, False, 42, mx, True)
        g.add_sequence(seq, True)
        if mx == h - 1:
            assert g.mins == [] and g.abunds == []
        else:
            assert g.mins == [h]
            # num mode: a full sketch would stop counting its maximum (Q3); with num=10 it is not full
            assert g.abunds == [n - k + 1]
    # mixed: the repeat sits between two random stretches
    rnd = bytes(coracle.synth_dna(0, 40000, 21, 0))
    seq2 = rnd[:20000] + b"ac" * 300000 + rnd[20000:]
    for case in [(0, k, False, 42, 1 << 63, True), (50, k, False, 42, 0, True)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        g.add_sequence(seq2, True); o.add_sequence(seq2, True)
        same_state(g, o)


def test_repeated_device_adds_accumulate(pkg, coracle):
    import torch
    host = bytes(coracle.synth_dna(0, 600000, 33, 0))
    buf = torch.frombuffer(bytearray(host), dtype=torch.uint8).cuda()
    mx = 1 << 57
    g, o = pkg.KmerMinHash(0, 31, False, 42, mx, True), coracle.MinHash(0, 31, False, 42, mx, True)
    g.add_sequences_dev(buf.data_ptr(), 300000, [0, 300000], True)          # state stays in HBM
    assert len(g) == len(set(g.mins))
    g.add_sequences_dev(buf.data_ptr() + 200000, 400000, [0, 150000, 400000], True)   # materialises, merges
    o.add_sequence(host[:300000], True)
    o.add_sequence(host[200000:350000], True)
    o.add_sequence(host[350000:600000], True)
    same_state(g, o)
    # a copy of a device-resident sketch is a host sketch with the same content
    g2 = pkg.KmerMinHash(0, 31, False, 42, mx, True)
    g2.add_sequences_dev(buf.data_ptr(), 300000, [0, 300000], True)
    from sourmash_rust_amd import signature as S
    sig = S.Signature()
    sig.push_mh(g2)
    o2 = coracle.MinHash(0, 31, False, 42, mx, True)
    o2.add_sequence(host[:300000], True)
    assert sig.first_mh().mins == o2.mins and g2.mins == o2.mins and g2.abunds == o2.abunds


def test_bulk_add_many_and_device_merge(pkg, coracle):
    """add_many over a large array goes through the device fold; merging two large sketches goes
    through the device union.  Both must equal the reference's one-by-one semantics."""
    rng = np.random.RandomState(17)
    universe = rng.randint(0, 1 << 62, size=60000, dtype=np.int64).astype(np.uint64)
    stream = rng.choice(universe, 250000)                     # many repeats
    for case in [(0, 21, False, 42, 1 << 61, True), (0, 21, False, 42, 1 << 61, False), (300, 21, False, 42, 0, True),
                 (300, 21, False, 42, 0, False), (50, 21, False, 42, 1 << 61, True)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        g.add_many(stream[:100]); o.add_many(stream[:100])    # scalar path first
        g.add_many(stream); o.add_many(stream)                # bulk path on a non-empty sketch
        same_state(g, o)
    # large merges: both tracked / none tracked (Q5: abundances become Some([]) and are never truncated)
    for track in (True, False):
        for num, mx in [(0, 1 << 62), (40000, 0)]:
            ga, oa = pkg.KmerMinHash(num, 21, False, 42, mx, track), coracle.MinHash(num, 21, False, 42, mx, track)
            gb, ob = pkg.KmerMinHash(num, 21, False, 42, mx, track), coracle.MinHash(num, 21, False, 42, mx, track)
            a = rng.choice(universe, 120000); b = rng.choice(universe, 90000)
            ga.add_many(a); oa.add_many(a)
            gb.add_many(b); ob.add_many(b)
            same_state(ga, oa); same_state(gb, ob)
            assert len(ga) + len(gb) >= (1 << 16)
            ga.merge(gb); oa.merge(ob)
            same_state(ga, oa)
            assert ga.track_abundance


def test_degenerate_ksize_panics_like_the_reference(pkg, coracle):
    # slice::windows(0) panics in the reference: DNA with ksize 0, protein with ksize < 3 (Q8)
    for case in [(10, 0, False, 42, 0, False), (10, 2, True, 42, 0, False), (10, 1, True, 42, 0, True)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        with pytest.raises(pkg.SourmashError) as eg:
            g.add_sequence(b"ACGTACGTAC", True)
        with pytest.raises(coracle.OracleError) as eo:
            o.add_sequence(b"ACGTACGTAC", True)
        assert eg.value.code == eo.value.code == 1
        same_state(g, o)
    # shorter than ksize: silently nothing (reference src/lib.rs:257), also for protein
    for case in [(10, 21, False, 42, 0, False), (10, 21, True, 42, 0, False)]:
        g = pkg.KmerMinHash(*case)
        g.add_sequence(b"ACGTACGTAC", False)
        assert g.mins == []


def test_concurrent_callers(pkg, coracle):
    """Distinct sketches used from distinct threads (the reference's threading contract): the
    engine serialises device work, the error slot is per thread."""
    import threading
    seqs = [bytes(coracle.synth_dna(i * 100000, 60000, 40 + i, 0)) for i in range(6)]
    exp = []
    for s_ in seqs:
        o = coracle.MinHash(200, 21, False, 42, 0, True)
        o.add_sequence(s_, True)
        exp.append((o.mins, o.abunds))
    got, errs = [None] * 6, []

    def work(i):
        try:
            for _ in range(3):
                g = pkg.KmerMinHash(200, 21, False, 42, 0, True)
                g.add_sequence(seqs[i], True)
                got[i] = (g.mins, g.abunds, g.compare(g))
            if i % 2 == 0:   # an error on this thread must not leak to the others
                bad = pkg.KmerMinHash(5, 4)
                try:
                    bad.add_sequence(b"ACGTNACGT", False)
                except pkg.SourmashError as e:
                    assert e.code == 1101
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for i in range(6):
        assert got[i][0] == exp[i][0] and got[i][1] == exp[i][1] and got[i][2] == 1.0


def test_read_at_a_time_through_the_legacy_abi(pkg, coracle):
    """Thousands of small add_sequence calls are queued and hashed in device batches; the result and
    every error must be what one call at a time gives, whatever is interleaved with them."""
    import time
    rng = random.Random(99)
    for case in [(500, 21, False, 42, 0, True), (0, 31, False, 42, 1 << 58, True), (40, 15, True, 42, 0, True),
                 (6, 5, False, 7, 1 << 62, True)]:
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        for i in range(1500):
            r = rand_seq(rng, rng.choice([0, 10, 40, 151, 151, 151, 600]), bad=rng.choice([0, 0, 0, 0.01]))
            force = rng.random() < 0.5
            eg = eo = None
            try:
                g.add_sequence(r, force)
            except pkg.SourmashError as e:
                eg = (e.code, e.message.split(": ")[-1])
            try:
                o.add_sequence(r, force)
            except coracle.OracleError as e:
                eo = (e.code, e.message)
            assert eg == eo
            if i % 397 == 0:
                h = rng.getrandbits(60)
                g.add_hash(h); o.add_hash(h)          # scalar op in the middle: order must hold
            if i % 501 == 0:
                assert len(g) == len(o.mins)
        same_state(g, o)
    # and it is fast: 20 000 reads of 150 bp
    reads = [rand_seq(rng, 150, bad=0.0) for _ in range(20000)]
    g = pkg.KmerMinHash(0, 31, False, 42, (1 << 64) // 1000, False)
    t0 = time.perf_counter()
    for r in reads:
        g.add_sequence(r, True)
    n = len(g)
    dt = time.perf_counter() - t0
    o = coracle.MinHash(0, 31, False, 42, (1 << 64) // 1000, False)
    for r in reads:
        o.add_sequence(r, True)
    assert g.mins == o.mins and n == len(o.mins)
    assert dt < 2.0, "per-read calls must not pay a device launch each (took %.2f s)" % dt


def _grouped_case(pkg, coracle, cases, recs, groups, force, prefill=None):
    gs = [pkg.KmerMinHash(*c) for c in cases]
    os_ = [coracle.MinHash(*c) for c in cases]
    if prefill:
        for g, o in zip(gs, os_):
            g.add_sequence(prefill, True); o.add_sequence(prefill, True)
    first = None
    for r, grp in zip(recs, groups):
        try:
            os_[grp].add_sequence(r, force)
        except coracle.OracleError as e:
            first = first or e.message
    if first is None:
        pkg.KmerMinHash.add_sequences_grouped(gs, recs, groups, force)
    else:
        with pytest.raises(pkg.SourmashError) as ei:
            pkg.KmerMinHash.add_sequences_grouped(gs, recs, groups, force)
        assert ei.value.code == 1101 and ei.value.message.endswith(first)
    for g, o in zip(gs, os_):
        same_state(g, o)


def test_grouped_sketching(pkg, coracle):
    """smh_add_sequences_grouped: record r feeds sketches[groups[r]].  One launch + one (group, hash)
    sort for scaled DNA sketches with equal parameters; sketch-by-sketch service for the rest.  Each
    sketch must equal the oracle fed its own records in order."""
    rng = random.Random(77)
    n_groups = 23
    recs = [rand_seq(rng, rng.choice([0, 5, 20, 21, 22, 100, 151, 2000, 30000]), bad=rng.choice([0, 0, 0.002]))
            for _ in range(400)]
    interleaved = [rng.randrange(n_groups) for _ in recs]
    contiguous = sorted(interleaved)
    scaled = (0, 21, False, 42, 1 << 58, True)
    for groups in (interleaved, contiguous):
        _grouped_case(pkg, coracle, [scaled] * n_groups, recs, groups, True)                      # shared launch
        _grouped_case(pkg, coracle, [scaled[:5] + (False,)] * n_groups, recs, groups, True, prefill=recs[-1])
        _grouped_case(pkg, coracle, [scaled] * n_groups, recs, groups, False)                     # first error reported
        # tracking differs per sketch: still one launch
        _grouped_case(pkg, coracle, [scaled[:5] + (g % 2 == 0,) for g in range(n_groups)], recs, groups, True)
    # bottom-num sketches without abundance: per-group thresholds, one launch per run of records,
    # repetitive groups (fewer than num distinct hashes under the threshold) re-served on their own
    rep = list(recs)
    rep[5] = b"ACGTTGCA" * 6000
    rep[9] = b"A" * 40000
    rep[11] = rand_seq(rng, 300, 0) * 200
    for groups in (interleaved, contiguous):
        _grouped_case(pkg, coracle, [(50, 21, False, 42, 0, False)] * n_groups, rep, groups, True)
        _grouped_case(pkg, coracle, [(3 + 40 * g, 21, False, 42, 0, False) for g in range(n_groups)], rep, groups, True,
                      prefill=recs[-2])
        _grouped_case(pkg, coracle, [(500, 16, False, 7, 0, False)] * n_groups, rep, groups, False)
        # with abundance tracking the stream positions ride along (quirk Q3: the last element of a full sketch)
        _grouped_case(pkg, coracle, [(50, 21, False, 42, 0, True)] * n_groups, rep, groups, True)
        _grouped_case(pkg, coracle, [(7 + 30 * g, 21, False, 42, 0, g % 2 == 0) for g in range(n_groups)], rep, groups, True,
                      prefill=recs[-2])
    # parameter combinations the shared launch does not serve
    _grouped_case(pkg, coracle, [(50, 21, False, 42, 0, True)] * n_groups, recs, interleaved, True)
    _grouped_case(pkg, coracle, [(0, 21, False, 42, (1 << 58) + g, False) for g in range(n_groups)], recs, contiguous, True)
    _grouped_case(pkg, coracle, [(0, 21, True, 42, 1 << 60, True)] * n_groups, recs[:60], interleaved[:60], True)
    _grouped_case(pkg, coracle, [(5, 5, False, 42, 1 << 62, True)] * n_groups, recs[:100], interleaved[:100], True)
    # a group without records stays empty; group ids out of range are rejected
    gs = [pkg.KmerMinHash(*scaled) for _ in range(3)]
    pkg.KmerMinHash.add_sequences_grouped(gs, [rand_seq(rng, 5000, 0), rand_seq(rng, 5000, 0)], [2, 0], True)
    assert gs[1].mins == [] and gs[0].mins and gs[2].mins
    with pytest.raises(pkg.SourmashError) as ei:
        pkg.KmerMinHash.add_sequences_grouped(gs, [b"ACGT" * 50], [3], True)
    assert ei.value.code == 2
    # the same sketch listed for two groups receives both
    a = pkg.KmerMinHash(*scaled); o = coracle.MinHash(*scaled)
    pkg.KmerMinHash.add_sequences_grouped([a, a], [recs[3], recs[5], recs[8]], [0, 1, 0], True)
    for r in (recs[3], recs[5], recs[8]):
        o.add_sequence(r, True)
    same_state(a, o)


def test_large_host_input_is_pipelined_and_identical(pkg):
    """Host bytes of 256 MB and more (scaled DNA, force=true) are uploaded in chunks on a second stream
    while earlier chunks are hashed.  Same sketch as the device-resident path, with records that do
    not line up with the chunks, invalid bytes, and a second call that merges into existing state."""
    import ctypes as C
    import torch
    L = pkg.lib()
    n = 300_000_017
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, n, 12, 99991, None) == 0
    torch.cuda.synchronize()
    host = buf[:n].cpu().numpy()
    u64p = C.POINTER(C.c_uint64)
    for nrec, track in ((1, False), (977, True)):
        off = (np.arange(nrec + 1, dtype=np.uint64) * np.uint64(n // nrec)); off[-1] = n
        a = pkg.KmerMinHash(0, 31, False, 42, (1 << 64) // 500, track)
        b = pkg.KmerMinHash(0, 31, False, 42, (1 << 64) // 500, track)
        for rep in range(2):                       # the second pass merges into a non-empty sketch
            assert L.smh_add_sequences(a._p, host.ctypes.data_as(C.c_char_p), off.ctypes.data_as(u64p), nrec, True) == 0
            b.add_sequences_dev(buf.data_ptr(), n, off, True)
            assert np.array_equal(a.mins_np(), b.mins_np())
            if track:
                assert np.array_equal(a.abunds_np(), b.abunds_np())
        assert len(a) > 500_000


def test_release_workspace_then_continue(pkg, coracle):
    """smh_release_workspace frees the grow-only device buffers; the next calls re-create them."""
    rng = random.Random(4)
    seq = rand_seq(rng, 200000, 0)
    L = pkg.lib()
    for case in ((0, 21, False, 42, 1 << 60, True), (100, 27, True, 42, 0, False)):
        g, o = pkg.KmerMinHash(*case), coracle.MinHash(*case)
        g.add_sequence(seq, True); o.add_sequence(seq, True)
        assert g.compare(g) == 1.0
        assert L.smh_release_workspace() == 0
        g.add_sequence(seq[::-1], True); o.add_sequence(seq[::-1], True)
        same_state(g, o)
        assert g.compare(g) == o.compare(o)
        out = pkg.matrix.compare_block([g] * 20, [g] * 70, want=("jaccard",))
        assert (out["jaccard"] == 1.0).all()
        assert L.smh_release_workspace() == 0


@pytest.mark.parametrize("prot", [False, True])
def test_a_batch_of_many_short_records(prot, pkg, coracle):
    """100 000 records of 0..70 bases in one device batch (a batch of reads; above 65 536 records the device counts
    the records of at least ksize bases and the protein arm's positions itself) against the C oracle fed record by
    record: scaled sketch with abundance, and a bottom-num one."""
    import torch
    rng = np.random.default_rng(77)
    nrec = 100_000
    lens = rng.integers(0, 71, size=nrec)
    off = np.zeros(nrec + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    total = int(off[-1])
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=total)
    seq[rng.integers(0, total, size=total // 400)] = ord("N")
    buf = torch.from_numpy(seq).cuda()
    for num, mx in ((0, (1 << 64) // 50), (300, 0)):
        ks = 27 if prot else 21
        g = pkg.KmerMinHash(num, ks, prot, 42, mx, True)
        g.add_sequences_dev(buf.data_ptr(), total, off, True)
        o = coracle.MinHash(num, ks, prot, 42, mx, True)
        raw = seq.tobytes()
        for r in range(nrec):
            o.add_sequence(raw[int(off[r]):int(off[r + 1])], True)
        same_state(g, o)


def _count(pkg, name):
    import ctypes as C
    ms, k = C.c_double(), C.c_uint64()
    pkg.lib().smh_profile_get(name.encode(), C.byref(ms), C.byref(k))
    return k.value


@pytest.mark.parametrize("protein,track", [(False, True), (False, False), (True, True)])
def test_scaled_sketch_accumulates_in_hbm_small(protein, track, pkg, coracle):
    """Many add_sequence batches into ONE scaled sketch (the reference's normal use, src/lib.rs:252-305): after the first
    batch the state lives in HBM and every later batch is united with it THERE (sort.hip sorted_union_async) -- equal to the
    oracle fed the same records in the same order, abundances included, with batches that repeat earlier k-mers, an empty
    batch (records shorter than k), a batch that adds nothing new, and the legacy one-string calls in between."""
    mx = (1 << 64) // 50
    ks = 27 if protein else 21
    g = pkg.KmerMinHash(0, ks, protein, 42, mx, track)
    o = coracle.MinHash(0, ks, protein, 42, mx, track)
    L = pkg.lib()
    L.smh_profile_reset()
    recs = [bytes(coracle.synth_dna(i * 40000, 40000, 9, 0)) for i in range(12)]
    batches = [recs[0:3], recs[3:4], [b"ACGT"], recs[2:6], recs[6:12], recs[0:1], recs[11:12]]
    for bi, batch in enumerate(batches):
        g.add_sequences(batch, True)
        for r in batch:
            o.add_sequence(r, True)
        if bi == 3:                      # the one-string ABI in between (queued, drained by the next batch)
            g.add_sequence(recs[7][:5000], True); o.add_sequence(recs[7][:5000], True)
        assert len(g) == len(o.mins)
    assert _count(pkg, "sketch_to_host") == 0, "the sketch left HBM between batches"
    assert _count(pkg, "sketch_union_on_device") >= 5
    h = pkg.KmerMinHash(0, ks, protein, 42, mx, track)
    h.add_sequences(recs, True)
    assert g.compare(h) == 1.0 and g.count_common(h) == len(o.mins)          # compared where they are: still nothing copied
    assert _count(pkg, "sketch_to_host") == 0
    same_state(g, o)
    assert _count(pkg, "sketch_to_host") == 1
    g.add_sequences(recs[4:5], True); o.add_sequence(recs[4], True)           # a host-resident state keeps working (host merge)
    same_state(g, o)


def test_ten_batches_of_one_gb_into_one_sketch(pkg, coracle):
    """10 x 1 GB (1 000 records x 1 MB each) into ONE scaled sketch with abundances == the sketch of the 10 GB in one
    call; nothing is copied to the host before an accessor asks (profile counter), a compare against the one-shot sketch
    does not materialise either of them, sampled records sketched by the C oracle are contained in it, and a later batch
    costs about what the first one did (the union is rank arithmetic + two scatters, not a sort of the whole state)."""
    import ctypes as C
    import time
    import torch
    nrec, rlen, mx = 10000, 1_000_000, 18446744073709552
    total = nrec * rlen
    buf = torch.empty(total, dtype=torch.uint8, device="cuda")
    L = pkg.lib()
    assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), 0, total, 2, 0, C.c_void_p(0)) == 0
    torch.cuda.synchronize()
    off = np.arange(nrec + 1, dtype=np.uint64) * np.uint64(rlen)

    def add(mh, r0, r1):
        mh.add_sequences_dev(buf.data_ptr() + r0 * rlen, (r1 - r0) * rlen, off[r0:r1 + 1] - off[r0], True)

    for track in (True, False):
        whole = pkg.KmerMinHash(0, 31, False, 42, mx, track)
        add(whole, 0, nrec)
        for rep in range(2):             # the second pass is the timed one (buffers come from the pool by then)
            acc = pkg.KmerMinHash(0, 31, False, 42, mx, track)
            L.smh_profile_reset()
            times = []
            for b in range(10):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                add(acc, b * 1000, (b + 1) * 1000)
                n_now = len(acc)
                times.append(time.perf_counter() - t0)
        assert _count(pkg, "sketch_to_host") == 0 and _count(pkg, "sketch_union_on_device") == 9
        assert n_now == len(whole) and acc.compare(whole) == 1.0 and acc.count_common(whole) == n_now
        assert _count(pkg, "sketch_to_host") == 0
        print("track=%s batch ms: %s" % (track, " ".join("%.2f" % (t * 1e3) for t in times)))
        assert max(times[1:]) <= 1.5 * times[0], times
        assert (acc.mins_np() == whole.mins_np()).all()
        if track:
            assert (acc.abunds_np() == whole.abunds_np()).all()
            wm = dict(zip(acc.mins_np().tolist(), acc.abunds_np().tolist()))
            for r in (0, 5500, nrec - 1):
                o = coracle.MinHash(0, 31, False, 42, mx, True)
                o.add_sequence(bytes(coracle.synth_dna(r * rlen, rlen, 2, 0)), True)
                assert all(wm.get(h, 0) >= c for h, c in zip(o.mins, o.abunds))


@pytest.mark.parametrize("protein,track", [(False, True), (True, True), (False, False)])
def test_union_of_partial_sketches_on_the_device(protein, track, pkg, coracle):
    """Row e2 of SURVEY.md 8e: the per-rank partial sketches of one input folded into one sketch without leaving HBM
    (smh_sketch_export_dev / smh_sketch_absorb_dev; distributed.union_across_ranks with the other ranks' sketches passed in).
    Scaled sketches: the union is exact, abundances add (KmerMinHash::merge, reference src/lib.rs:307-403) -- against the
    oracle fed all records, with overlapping shards (the same k-mers on several ranks), an EMPTY rank and a rank whose
    state had already been brought to the host."""
    from sourmash_rust_amd import distributed as D
    mx = (1 << 64) // 40
    ks = 27 if protein else 21
    recs = [bytes(coracle.synth_dna(i * 30000, 30000, 13, 0)) for i in range(10)]
    shards = [recs[0:4], recs[3:7], [], recs[6:10], recs[0:1]]
    parts = []
    o = coracle.MinHash(0, ks, protein, 42, mx, track)
    for sh in shards:
        p = pkg.KmerMinHash(0, ks, protein, 42, mx, track)
        if sh:
            p.add_sequences(sh, True)
        for r in sh:
            o.add_sequence(r, True)
        parts.append(p)
    _ = parts[3].mins                         # this rank's sketch was looked at: its state is on the host now
    pkg.lib().smh_profile_reset()
    uni = D.union_across_ranks(parts[0], parts=parts)
    assert _count(pkg, "sketch_to_host") == 0
    same_state(uni, o)
    for p, sh in zip(parts, shards):          # the parts themselves are unchanged
        q = coracle.MinHash(0, ks, protein, 42, mx, track)
        for r in sh:
            q.add_sequence(r, True)
        same_state(p, q)


@pytest.mark.parametrize("params", [(0, 9, False, 42, 1 << 61, True), (300, 9, False, 42, 0, True), (50, 6, False, 7, 0, False)])
def test_add_word_calls_are_queued_and_keep_their_order(params, pkg, coracle):
    """kmerminhash_add_word (reference src/ffi.rs:82-95, src/lib.rs:247-250) through the legacy one-word-per-call ABI: the
    words are hashed by one device launch when the sketch is next observed, then go through add_hash in call order --
    interleaved with add_hash, add_sequence and accessors, in scaled and in bottom-num mode (where the abundance of the
    last element depends on the order, quirk Q3), with repeated words and the empty word."""
    import time
    rng = random.Random(5)
    g, o = pkg.KmerMinHash(*params), coracle.MinHash(*params)
    alphabet = [bytes(rng.choice(b"ACGT") for _ in range(params[1])) for _ in range(400)] + [b""]
    for step in range(3000):
        r = rng.random()
        if r < 0.90:
            w = rng.choice(alphabet)
            g.add_word(w); o.add_word(w)
        elif r < 0.95:
            h = rng.getrandbits(60)
            g.add_hash(h); o.add_hash(h)
        elif r < 0.98:
            s = bytes(rng.choice(b"ACGT") for _ in range(60))
            g.add_sequence(s, True); o.add_sequence(s, True)
        else:
            assert len(g) == len(o.mins)
    same_state(g, o)
    # 100 000 calls: one launch per 64 K words instead of one per word
    g2 = pkg.KmerMinHash(0, 9, False, 42, 1 << 62, True)
    words = [rng.choice(alphabet[:-1]) for _ in range(100000)]
    t0 = time.perf_counter()
    for w in words:
        g2.add_word(w)
    n = len(g2)
    dt = time.perf_counter() - t0
    o2 = coracle.MinHash(0, 9, False, 42, 1 << 62, True)
    for w in set(words):
        o2.add_word(w)
    assert n == len(o2.mins) and g2.mins == o2.mins and sum(g2.abunds) == sum(1 for w in words if coracle.hash_murmur(w, 42) <= (1 << 62))
    print("100 000 add_word calls: %.3f s" % dt)
    assert dt < 1.0
