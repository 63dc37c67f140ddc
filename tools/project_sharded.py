"""What each rank of a `world`-rank job spends in its compute phases of the sharded 10 000 x 10 000 matrix, measured on ONE GPU by
playing the ranks one after the other (the collectives are not measured: there is one GPU here): dictionary slice
(smh_collection_begin), dictionary assembly (smh_collection_finish), block compare with pair ownership (smh_collection_compare),
the local part of the mirror exchange (transposes + selects).  max over ranks per phase = the job's critical path.
    python tools/project_sharded.py [N] [collection: families|one_family] [worlds...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package()
from sourmash_rust_amd import distributed as D, synth, matrix as MX

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
kind = sys.argv[2] if len(sys.argv) > 2 else "one_family"
worlds = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8]
num = 2000
sigs = synth.family_signatures(0, n, num=num, seed=3, n_families=1 if kind == "one_family" else 50)
t = torch.from_numpy(sigs.view(np.int64)).cuda()
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(num)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) * 1e3


import contextlib
_lim = os.environ.get("PROF_COMP_LIMIT")
_ctx = MX.tuning(comp_pairs_limit=int(_lim)) if _lim else contextlib.nullcontext()
_ctx.__enter__()
for world in worlds:
    blocks = [D.shard_range(n, world, r)[:2] for r in range(world)]
    best = None
    for rep in range(3):
        ph = {"slice": [], "assemble": [], "compare": [], "exchange_local": []}
        colls = []
        for r in range(world):
            c, ms = timed(lambda: MX.Collection(t, off, world, r))
            colls.append(c); ph["slice"].append(ms)
        gathered = None
        if world > 1:
            gathered = torch.empty(world * colls[0].share_bytes, dtype=torch.uint8, device="cuda")
            for r, c in enumerate(colls):
                c.share_to(gathered[r * c.share_bytes:(r + 1) * c.share_bytes])
        outs = []
        for r, c in enumerate(colls):
            _, ms = timed(lambda: c.finish(gathered)); ph["assemble"].append(ms)
            own = MX.OWN_CIRCULAR if world > 1 else MX.OWN_TRIANGLE
            o, ms = timed(lambda: c.compare(blocks[r][0], blocks[r][1], num, want=("jaccard",), ownership=own)); ph["compare"].append(ms)
            if r == 0:
                st0 = MX.last_stats()
            outs.append(o["jaccard"])
        if world > 1:
            sends = []
            for r in range(world):
                s, ms = timed(lambda: D.mirror_send_list(outs[r], blocks, r, n)); sends.append(s); ph["exchange_local"].append(ms)
            for r in range(world):
                recv = [sends[p][r] for p in range(world)]
                _, ms = timed(lambda: D.mirror_apply(outs[r], recv, blocks, r, n)); ph["exchange_local"][r] += ms
        else:
            ph["exchange_local"] = [0.0]
        for c in colls:
            c.close()
        tot = sum(max(v) for v in ph.values())
        if best is None or tot < best[0]:
            best = (tot, {k: (max(v), sum(v) / len(v)) for k, v in ph.items()}, colls[0].share_bytes,
                    sum(s.numel() * 8 for s in (sends[0] if world > 1 else [])))
        del outs, gathered
    tot, ph, share, sent = best
    print("N=%d %s world=%d: compute critical path %.2f ms | " % (n, kind, world, tot) +
          " ".join("%s max %.2f avg %.2f" % (k, v[0], v[1]) for k, v in ph.items()) +
          " | share %.1f MB per rank, mirrored blocks sent by rank 0: %.1f MB | rank 0: %s, %d of %d tiles of %d pairs" %
          (share / 1e6, sent / 1e6, st0["route"], st0["tiles_visited"], st0["tiles_total"], st0["pairs_per_tile"]), flush=True)
