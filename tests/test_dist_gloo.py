"""world_size-2 CPU test (gloo) of the N>1 path: the same sharding + all-gather code that bench.py
runs over RCCL, with the block compute done by the oracle (there is no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import coracle
    from __graft_entry__ import load_package
    load_package()
    from sourmash_rust_amd import distributed as D, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    num = 200
    lo, hi, per = D.shard_range(n_total, world, rank)
    local = np.zeros((per, num), dtype=np.uint64)
    local[: hi - lo] = synth.family_signatures(lo, hi, num=num, n_families=3, pool=300, private=60, seed=5)
    local_t = torch.from_numpy(local.view(np.int64))

    def oracle_block(rows_t, n_rows, cols_t, n_cols, num_, want):
        rows = [rows_t[i].numpy().view(np.uint64) for i in range(n_rows)]
        cols = [cols_t[j].numpy().view(np.uint64) for j in range(n_cols)]
        common, size, jac = coracle.compare_matrix(rows, cols, num_, 31, 0)
        return {"jaccard": torch.from_numpy(jac), "common": torch.from_numpy(common.view(np.int64))}

    out = D.compare_matrix_sharded(local_t, n_total, num, want=("jaccard", "common"), compute_block=oracle_block)

    # the sketch side: every rank sketches its records, then the optional union
    recs_lo, recs_hi = D.shard_records(7, world, rank)
    mh = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
    for r in range(recs_lo, recs_hi):
        mh.add_sequence(bytes(coracle.synth_dna(r * 5000, 5000, 9, 0)), True)
    parts = D.merge_sketch_across_ranks(torch.from_numpy(mh.mins_np().view(np.int64)),
                                        torch.from_numpy(mh.abunds_np().view(np.int64)))
    q.put((rank, lo, hi, out["jaccard"].numpy(), out["common"].numpy(),
           [(m.numpy().view(np.uint64), a.numpy().view(np.uint64)) for m, a in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [9, 10])
def test_row_sharded_matrix_and_sketch_union_gloo(n_total, coracle):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from sourmash_rust_amd import synth

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    sigs = synth.family_signatures(0, n_total, num=200, n_families=3, pool=300, private=60, seed=5)
    rows = [sigs[i] for i in range(n_total)]
    common, size, jac = coracle.compare_matrix(rows, rows, 200, 31, 0)
    got_j = np.concatenate([r[3] for r in res], axis=0)
    got_c = np.concatenate([r[4] for r in res], axis=0)
    assert [(r[1], r[2]) for r in res] == [(0, (n_total + 1) // 2), ((n_total + 1) // 2, n_total)]
    assert (got_j == jac).all() and (got_c.view(np.uint64) == common).all()

    # union of the per-rank scaled sketches == sketch of all records on one rank (exact, abundances add)
    whole = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
    for r in range(7):
        whole.add_sequence(bytes(coracle.synth_dna(r * 5000, 5000, 9, 0)), True)
    for rank_res in res:
        merged = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
        for m, a in rank_res[5]:
            part = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
            for h, c in zip(m, a):
                part.mins_push(int(h)); part.abunds_push(int(c))
            merged.merge(part)
        assert merged.mins == whole.mins and merged.abunds == whole.abunds
