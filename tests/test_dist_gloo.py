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


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (the shape of the driver's command, no torchrun around it) must
    start two ranks by itself.  The children see NO GPU (HIP/CUDA_VISIBLE_DEVICES are emptied before
    anything is launched, so this behaves the same on a CPU box, a 1-GPU box and a GPU node): each
    rank stops at its first line of GPU set-up -- which proves the launch went through: both ranks
    ran, with RANK/WORLD_SIZE set.  The real 2-rank run is tests/test_gpu_multirank.py."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    text = r.stdout + r.stderr
    assert "rank 0 needs cuda:0" in text and "rank 1 needs cuda:1" in text


def test_bench_cpu_workers(tmp_path, coracle):
    """The CPU-baseline workers of bench.py (`--cpu-worker`): bounded samples through the C oracle."""
    import json
    import subprocess

    def run(*argv):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker"] + [str(a) for a in argv],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])

    d = run("config0", tmp_path / "c0.npy")
    assert d["self_compare"] == 1.0 and d["kmers"] == 999970 and len(np.load(tmp_path / "c0.npy")) == 500
    d = run("sketch", tmp_path / "s.npy", 3, 5, 0.01)
    assert d["records"] == 1
    o = coracle.MinHash(0, 31, False, 42, 18446744073709552, False)
    o.add_sequence(bytes(coracle.synth_dna(3 * 1000000, 1000000, 2, 0)), True)
    assert (np.load(tmp_path / "s.npy") == o.mins_np()).all()
    d = run("compare", tmp_path / "c.npy", 60, 2, 60, 0.05, 1)
    j = np.load(tmp_path / "c.npy")
    assert j.shape == (d["rows"], 60) and j[0, 2] == 1.0 and (j > 0).all()      # contaminant: every pair shares a hash
