"""include/*.h compile as plain C and a C program linked against libsourmash_amd.so drives the ABI."""
import os
import subprocess

from conftest import ROOT


def test_c_client_builds_and_runs(pkg, tmp_path):
    libdir = os.path.dirname(pkg.SO_PATH)
    exe = str(tmp_path / "c_abi_client")
    subprocess.check_call(["gcc", "-std=c11", "-D_GNU_SOURCE", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi_client.c"), "-o", exe,
                           "-L", libdir, "-lsourmash_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe], text=True, stderr=subprocess.STDOUT)
    assert "c abi client ok" in out


def test_oracle_kats_under_sanitizers(tmp_path):
    """The C oracle's own self-test (reference KATs) under ASan + UBSan -- sanitizers run on the CPU
    build only (GPU ASan is not available on this pool)."""
    exe = str(tmp_path / "oracle_selftest")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "oracle", "selftest.c"), os.path.join(ROOT, "oracle", "sourmash_oracle.c"),
                           "-I", os.path.join(ROOT, "oracle"), "-o", exe])
    out = subprocess.check_output([exe], text=True, stderr=subprocess.STDOUT)
    assert "oracle selftest ok" in out
