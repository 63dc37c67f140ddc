"""pytest configuration.

Markers: `gpu` = needs a real MI355X (run with `-m gpu` on the GPU box).  Everything else
runs on CPU.  The oracle (oracle/) is test infrastructure: only tests import it.
"""
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X GPU")
    # a hung test must fail with a traceback of where it hangs, not stall the whole run in silence
    # (pytest-timeout is part of the image; without it the option simply does not exist)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 840


@pytest.fixture(scope="session")
def coracle():
    import coracle as m
    m.build()
    return m


@pytest.fixture(scope="session")
def pyoracle():
    import pyoracle as m
    return m


def _load_sig_file(path):
    with open(path) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def sbt_v5_leaves():
    """{leaf position: first sketch dict} of the reference fixture tests/data/v5.sbt.json."""
    tree = _load_sig_file(os.path.join(GOLDEN, "v5.sbt.json"))
    out = {}
    for pos, leaf in tree["leaves"].items():
        sig = _load_sig_file(os.path.join(GOLDEN, "sbt_v5", leaf["filename"] + ".sig"))
        out[int(pos)] = sig[0]["signatures"][0]
    return out


@pytest.fixture(scope="session")
def sbt_subset_sketches():
    """First sketch of each of the 100 leaf signatures of tests/data/.sbt.subset (k=21, scaled, abund)."""
    with gzip.open(os.path.join(GOLDEN, "sbt_subset_sigs.json.gz"), "rt") as fh:
        d = json.load(fh)
    return [sorted_sketch(d[k][0]["signatures"][0]) for k in sorted(d)]


def sorted_sketch(sk):
    """The .sbt.subset fixture files store `mins` in arbitrary order (written by an old Python
    sourmash); KmerMinHash's invariant is ascending mins, which is what every
    comparison assumes, so the fixture is put in that order (abundances permuted alike)."""
    order = sorted(range(len(sk["mins"])), key=lambda i: sk["mins"][i])
    out = dict(sk)
    out["mins"] = [sk["mins"][i] for i in order]
    if "abundances" in sk:
        out["abundances"] = [sk["abundances"][i] for i in order]
    return out


@pytest.fixture(scope="session")
def pkg():
    """The product package (sourmash-rust_amd/).  Built in-tree on first use (hipcc cross-compiles
    gfx950 without a GPU); on the GPU box the prebuilt .so travels with the snapshot."""
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "sourmash-rust_amd", "lib", "libsourmash_amd.so")):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def pkg_lib(pkg):
    """The loaded C-ABI library (ctypes); loading needs no GPU."""
    return pkg.lib()
