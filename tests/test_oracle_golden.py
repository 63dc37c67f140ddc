"""The C oracle reproduces every committed golden vector (tests/golden/make_golden.py)."""
import gzip
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


def golden():
    with gzip.open(os.path.join(GOLDEN, "golden_vectors.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.mark.parametrize("name", sorted(golden()["sketch"]))
def test_oracle_sketch_golden(name, coracle):
    g = golden()["sketch"][name]
    mh = coracle.MinHash(*g["params"])
    mh.add_sequence(bytes(coracle.synth_dna(*g["synth"])), g["force"])
    assert mh.mins == g["mins"] and mh.abunds == g["abunds"]


def test_oracle_matrix_golden(coracle, sbt_v5_leaves, sbt_subset_sketches):
    mats = np.load(os.path.join(GOLDEN, "golden_matrices.npz"))
    for tag, sks in (("v5", [sbt_v5_leaves[k] for k in sorted(sbt_v5_leaves)]), ("subset", sbt_subset_sketches)):
        num = 0 if sks[0]["max_hash"] else sks[0]["num"]
        arrs = [np.array(s["mins"], dtype=np.uint64) for s in sks]
        sub = arrs[:12]
        common, size, jac = coracle.compare_matrix(sub, arrs, num, sks[0]["ksize"], sks[0]["max_hash"])
        assert (common == mats[tag + "_common"][:12]).all()
        assert (size == mats[tag + "_size"][:12]).all()
        assert (jac == mats[tag + "_jaccard"][:12]).all()
