"""include/*.h compile as plain C and a C program linked against libsourmash_amd.so drives the ABI."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


# (tests/test_sanitizers.py links the clients against the ASan/UBSan build of the library, whose runtime is preloaded)
EXTRA_LD = os.environ.get("SMH_TEST_EXTRA_LDFLAGS", "").split()


def test_c_client_builds_and_runs(pkg, tmp_path):
    libdir = os.path.dirname(pkg.SO_PATH)
    exe = str(tmp_path / "c_abi_client")
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe,
                           "-L", libdir, "-lsourmash_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"] + EXTRA_LD)
    out = subprocess.check_output([exe], text=True, stderr=subprocess.STDOUT)
    assert "c abi client ok" in out


def _build_and_run(pkg, tmp_path, name):
    libdir = os.path.dirname(pkg.SO_PATH)
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", name + ".c"), "-o", exe,
                           "-L", libdir, "-lsourmash_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"] + EXTRA_LD)
    return subprocess.check_output([exe], text=True, stderr=subprocess.STDOUT)


def test_rust_shim_forwards_are_link_checked(pkg, tmp_path):
    """No Rust toolchain here: every symbol of the shim's extern block must (a) be exported by the
    library with the header's prototype -- the C program that calls each of them the way the shim
    does compiles with -Werror against include/*.h and links -- and (b) actually appear in that program."""
    shim = open(os.path.join(ROOT, "sourmash-rust_amd", "rust", "src", "lib.rs")).read()
    bound = set(re.findall(r"pub fn ((?:kmerminhash|sourmash|smh|signature|hash)_[a-z0-9_]+)\(", shim))
    assert len(bound) >= 30
    ctext = open(os.path.join(ROOT, "tests", "c_shim_symbols.c")).read()
    missing = sorted(sym for sym in bound if not re.search(r"\b%s\b" % sym, ctext))
    assert not missing, "symbols bound by the Rust shim that the C link-check does not call: %s" % missing
    exported = set(pkg.exported_symbols())
    assert bound <= exported, sorted(bound - exported)
    # every public method of the reference's impl block (src/lib.rs:141-513) exists in the shim, with its signature
    for sig in ("pub fn new(num: u32, ksize: u32, is_protein: bool, seed: u64, max_hash: u64, track_abundance: bool) -> KmerMinHash",
                "pub fn check_compatible(&self, other: &KmerMinHash) -> Result<bool, Error>",
                "pub fn add_hash(&mut self, hash: u64)", "pub fn add_word(&mut self, word: &[u8])",
                "pub fn add_sequence(&mut self, seq: &[u8], force: bool) -> Result<(), Error>",
                "pub fn merge(&mut self, other: &KmerMinHash) -> Result<(), Error>",
                "pub fn add_from(&mut self, other: &KmerMinHash) -> Result<(), Error>",
                "pub fn add_many(&mut self, hashes: &[u64]) -> Result<(), Error>",
                "pub fn add_many_with_abund(&mut self, hashes: &[(u64, u64)]) -> Result<(), Error>",
                "pub fn count_common(&self, other: &KmerMinHash) -> Result<u64, Error>",
                "pub fn intersection(&self, other: &KmerMinHash) -> Result<(Vec<u64>, u64), Error>",
                "pub fn intersection_size(&self, other: &KmerMinHash) -> Result<(u64, u64), Error>",
                "pub fn compare(&self, other: &KmerMinHash) -> Result<f64, Error>",
                "pub fn size(&self) -> usize", "pub fn _hash_murmur(kmer: &[u8], seed: u64) -> u64"):
        assert sig in shim, sig
    out = _build_and_run(pkg, tmp_path, "c_shim_symbols")
    assert "c shim symbols ok" in out


@pytest.mark.gpu
def test_rust_shim_link_check_on_the_gpu(pkg, tmp_path):
    assert "c shim symbols ok (gpu)" in _build_and_run(pkg, tmp_path, "c_shim_symbols")


@pytest.mark.gpu
def test_intersection_hashes_and_check_compatible(pkg, coracle):
    """KmerMinHash::intersection (reference src/lib.rs:438-468): the common hashes inside the bottom-num of
    the union + the combined size, against the oracle's two merges + two intersections."""
    import random
    rng = random.Random(12)
    for trial in range(40):
        num = rng.choice([0, 1, 5, 50, 2000])
        mx = 0 if num else 1 << 62
        uni = [rng.getrandbits(62) for _ in range(rng.choice([10, 200, 4000]))]
        ga, oa = pkg.KmerMinHash(num, 21, False, 42, mx), coracle.MinHash(num, 21, False, 42, mx)
        gb, ob = pkg.KmerMinHash(num, 21, False, 42, mx), coracle.MinHash(num, 21, False, 42, mx)
        for hsh in rng.choices(uni, k=rng.choice([0, 3, 100, 3000])):
            ga.add_hash(hsh); oa.add_hash(hsh)
        for hsh in rng.choices(uni, k=rng.choice([0, 3, 100, 3000])):
            gb.add_hash(hsh); ob.add_hash(hsh)
        common, size = ga.intersection_hashes(gb)
        oc, osz = oa.intersection_size(ob)
        assert (len(common), size) == (oc, osz)
        both = sorted(set(oa.mins) & set(ob.mins))
        assert common == both[:oc]
        assert ga.check_compatible(gb) is True
    with pytest.raises(pkg.SourmashError) as ei:
        pkg.KmerMinHash(5, 21).check_compatible(pkg.KmerMinHash(5, 31))
    assert ei.value.code == 101


def test_oracle_kats_under_sanitizers(tmp_path):
    """The C oracle's own self-test (reference KATs) under ASan + UBSan -- sanitizers run on the CPU
    build only (GPU ASan is not available on this pool)."""
    exe = str(tmp_path / "oracle_selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "oracle", "selftest.c"), os.path.join(ROOT, "oracle", "sourmash_oracle.c"),
                           "-I", os.path.join(ROOT, "oracle"), "-o", exe])
    out = subprocess.check_output([exe], text=True, stderr=subprocess.STDOUT)
    assert "oracle selftest ok" in out
