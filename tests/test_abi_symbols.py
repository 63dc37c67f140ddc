"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    names = set()
    for hdr in ("sourmash.h", "sourmash_amd.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        for m in re.finditer(r"\b([a-z_][a-z0-9_]*)\s*\(", text):
            n = m.group(1)
            if n.startswith(("kmerminhash_", "signature", "sourmash_", "smh_", "hash_murmur")):
                names.add(n)
    return names


def test_reference_abi_is_complete():
    # the 48 symbols of the reference header
    ref = [n for n in declared_symbols() if not n.startswith("smh_")]
    assert len(ref) == 48


def test_library_exports_every_declared_symbol(pkg):
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.SO_PATH], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = declared_symbols() - exported
    assert not missing, missing
    assert declared_symbols() == set(pkg.exported_symbols())


def test_loads_and_scalar_calls_work_without_gpu(pkg):
    L = pkg.lib()
    mh = pkg.KmerMinHash(3, 21, False, 42, 0, True)
    for h in (5, 3, 9, 3, 1):
        mh.add_hash(h)
    assert mh.mins == [1, 3, 5] and mh.abunds == [1, 2, 1]
    assert (mh.num, mh.ksize, mh.is_protein, mh.seed, mh.max_hash, mh.track_abundance) == (3, 21, False, 42, 0, True)
    assert L.sourmash_err_get_last_code() == 0


def test_compute_fails_loudly_without_gpu(pkg):
    if pkg.device_available():
        pytest.skip("a GPU is present")
    mh = pkg.KmerMinHash(20, 10)
    with pytest.raises(pkg.SourmashError) as ei:
        mh.add_sequence(b"TGCCGCCCAGCA")
    assert ei.value.code == 2 and "no HIP device" in ei.value.message
    assert len(mh) == 0
    with pytest.raises(pkg.SourmashError):
        pkg.hash_murmur(b"ACG")
    with pytest.raises(pkg.SourmashError):
        mh.compare(pkg.KmerMinHash(20, 10))
    # the additive entry points too: batch, grouped, block compare, resident index -- error code 2, no crash,
    # and nothing applied to the sketches
    a, b = pkg.KmerMinHash(0, 21, False, 42, 1 << 60), pkg.KmerMinHash(0, 21, False, 42, 1 << 60)
    for h in (7, 11, 13):
        a.add_hash(h); b.add_hash(h + 1)
    for call in (lambda: a.add_sequences([b"ACGT" * 20, b"TTGCA" * 10], True),
                 lambda: pkg.KmerMinHash.add_sequences_grouped([a, b], [b"ACGT" * 20, b"TTGCA" * 10], [0, 1], True),
                 lambda: pkg.matrix.compare_block([a, b], [a, b]),
                 lambda: pkg.index.ResidentIndex([a, b]),
                 lambda: pkg.index.search_minhashes([a, b], a, 0.1),
                 lambda: a.count_common(b)):
        with pytest.raises(pkg.SourmashError) as ei:
            call()
        assert ei.value.code == 2
    assert a.mins == [7, 11, 13] and b.mins == [8, 12, 14]
    # the diagnostic sort: code 2, the array untouched
    import ctypes as C
    import numpy as np
    keys = np.array([5, 3, 9], dtype=np.uint64)
    assert pkg.lib().smh_sort_u64(keys.ctypes.data_as(C.c_void_p), None, keys.size) == 2
    assert keys.tolist() == [5, 3, 9]


def test_product_does_not_reference_the_oracle():
    # the product path may not import, include, link or execute anything under oracle/
    pkgdir = os.path.join(ROOT, "sourmash-rust_amd")
    bad = re.compile(r"(import\s+(coracle|pyoracle)|from\s+oracle|#include\s+[\"<][^\">]*oracle|liboracle|"
                     r"oracle/(_build|_ref)|sourmash_oracle|omh_[a-z_]+\()")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not bad.search(text), os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", os.path.join(pkgdir, "lib", "libsourmash_amd.so")], text=True)
    assert "oracle" not in out
