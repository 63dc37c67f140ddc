import os, sys, json
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle"); sys.path.insert(0, "/root/repo/tests")
from __graft_entry__ import load_package
pkg = load_package()
import coracle
from conftest import GOLDEN
tree = json.load(open(os.path.join(GOLDEN, "v5.sbt.json")))
sks = []
for pos, leaf in sorted(tree["leaves"].items(), key=lambda kv: int(kv[0])):
    sig = json.load(open(os.path.join(GOLDEN, "sbt_v5", leaf["filename"] + ".sig")))
    sks.append(sig[0]["signatures"][0])
def mk(M, sk):
    mh = M(sk["num"], sk["ksize"], False, sk["seed"], sk["max_hash"], False)
    for m in sk["mins"]: mh.mins_push(m)
    return mh
g = [mk(pkg.KmerMinHash, s) for s in sks]; o = [mk(coracle.MinHash, s) for s in sks]
for masks in (True, False):
    with pkg.matrix.tuning(route="tiled", range_masks=masks):
        out = pkg.matrix.compare_block(g, g, want=("jaccard", "common", "size", "count_common"))
    print("masks", masks, pkg.matrix.last_stats())
    for i in range(len(g)):
        for j in range(len(g)):
            c, s_ = o[i].intersection_size(o[j]); cc = o[i].count_common(o[j])
            got = (int(out["common"][i, j]), int(out["size"][i, j]), int(out["count_common"][i, j]))
            if got != (c, s_, cc): print("  pair", i, j, "got", got, "want", (c, s_, cc), "lens", len(sks[i]["mins"]), len(sks[j]["mins"]), "num", sks[i]["num"])
