"""Pins the protein arm (SURVEY.md 8a rows `add_sequence` protein, `to_aa`, `CODONTABLE`) to DATA
taken from the reference: tests/golden/codontable.json holds the 64 (codon -> residue) pairs of the
reference's CODONTABLE literal (src/lib.rs:691-777) and the six-frame walk order of
src/lib.rs:280-300, extracted by tests/golden/make_codontable.py.  Here: both oracles against it
(CPU).  The device is checked against the same file in tests/test_gpu_sketch.py."""
import itertools
import json
import os
import random

import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def codon_doc():
    with open(os.path.join(GOLDEN, "codontable.json")) as fh:
        return json.load(fh)


_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def revcomp(s):
    return bytes(s).translate(_COMP)[::-1]


def table_translate(doc, seq):
    """to_aa (reference src/lib.rs:779-793) driven ONLY by the fixture's table: stop at the first
    incomplete codon, drop codons that are not among the 64."""
    t = doc["table"]
    out = bytearray()
    for i in range(0, len(seq) - 2, 3):
        aa = t.get(seq[i:i + 3].decode("latin-1"))
        if aa is not None:
            out.append(ord(aa))
    return bytes(out)


def table_windows(doc, seq, ksize):
    """every residue window of the protein arm, in the reference's frame order (fixture `frames`)"""
    seq = bytes(seq).upper()
    rc = revcomp(seq)
    aak = ksize // 3
    for strand, skip in doc["frames"]:
        src = seq if strand == "forward" else rc
        aa = table_translate(doc, src[skip:])
        for w in range(len(aa) - aak + 1):
            yield aa[w:w + aak]


def test_fixture_is_the_standard_code(codon_doc):
    t = codon_doc["table"]
    assert sorted(t) == sorted("".join(c) for c in itertools.product("ACGT", repeat=3))
    assert sum(1 for v in t.values() if v == "*") == 3 and t["ATG"] == "M" and t["TGG"] == "W"
    assert codon_doc["frames"] == [["forward", 0], ["revcomp", 0], ["forward", 1], ["revcomp", 1],
                                   ["forward", 2], ["revcomp", 2]]


def test_c_oracle_table(coracle, codon_doc):
    import ctypes as C
    L = coracle.lib()
    for codon, aa in codon_doc["table"].items():
        out = C.create_string_buffer(8)
        n = C.c_size_t()
        L.omh_translate_frames(codon.encode(), 3, 0, 0, out, C.byref(n))
        assert n.value == 1 and out.raw[:1] == aa.encode(), codon
        L.omh_translate_frames(codon.lower().encode(), 3, 0, 0, out, C.byref(n))   # upper-cased first (lib.rs:253-256)
        assert n.value == 1 and out.raw[:1] == aa.encode(), codon


def test_py_oracle_table(pyoracle, codon_doc):
    assert {k.decode(): chr(v) for k, v in pyoracle.CODONS.items()} == codon_doc["table"]
    for codon, aa in codon_doc["table"].items():
        assert pyoracle.to_aa(codon.encode()) == aa.encode()


@pytest.mark.parametrize("which", ["c", "py"])
def test_oracle_frames_and_windows(which, coracle, pyoracle, codon_doc):
    """add_sequence(protein) of each oracle == the fixture-driven window stream fed through that
    oracle's add_word, in num mode with abundance (where quirk Q3 makes the result depend on the
    ORDER of the stream, so the frame order is pinned too) and in scaled mode."""
    O = coracle if which == "c" else pyoracle
    rng = random.Random(5)
    for trial in range(12):
        n = rng.choice([9, 10, 11, 30, 200, 501])
        alphabet = b"ACGT" if trial % 3 else b"ACGTNacgtn"     # N: dropped codons (Q8); lower case
        seq = bytes(rng.choice(alphabet) for _ in range(n))
        for ksize, num, mx in ((9, 6, 0), (27, 10, 0), (21, 0, (1 << 64) // 3), (3, 4, 0)):
            a = O.MinHash(num, ksize, True, 42, mx, True)
            b = O.MinHash(num, ksize, True, 42, mx, True)
            a.add_sequence(seq, False)
            if len(seq) >= ksize:                                # lib.rs:257
                for w in table_windows(codon_doc, seq, ksize):
                    b.add_word(w)
            assert list(a.mins) == list(b.mins), (trial, ksize, num)
            assert list(a.abunds) == list(b.abunds), (trial, ksize, num)


def test_per_frame_translation_c_oracle(coracle, codon_doc):
    import ctypes as C
    L = coracle.lib()
    rng = random.Random(9)
    seq = bytes(rng.choice(b"ACGTN") for _ in range(300))
    rc = revcomp(seq)
    for strand, skip in codon_doc["frames"]:
        out = C.create_string_buffer(128)
        n = C.c_size_t()
        L.omh_translate_frames(seq, len(seq), skip, 1 if strand == "revcomp" else 0, out, C.byref(n))
        src = seq if strand == "forward" else rc
        assert out.raw[:n.value] == table_translate(codon_doc, src[skip:])
