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


class OracleEngine:
    """What computes a rank's row block in these CPU tests: the C oracle, pair by pair.  It honours the
    ownership contract of smh_collection_compare exactly as loosely as the contract allows: with
    ownership 2 every pair the rank's rows do NOT own (and whose column is not one of its own rows) is
    POISONED, so a test only passes if the exchange really delivers those entries from their owners."""

    def __init__(self, coracle, num_k=31):
        self.coracle = coracle

    def begin(self, allsigs, n_total, world, rank):
        self.sigs = [allsigs[i].numpy().view(np.uint64) for i in range(n_total)]
        self.n, self.world, self.rank = n_total, world, rank

    def share(self):
        import torch
        return torch.full((16,), self.rank, dtype=torch.uint8)

    def finish(self, gathered):
        assert (gathered is None) == (self.world == 1)
        if gathered is not None:       # rank-major concatenation of the shares
            assert gathered.reshape(self.world, 16)[:, 0].tolist() == list(range(self.world))

    def compare(self, lo, hi, num, want, ownership):
        import torch
        from sourmash_rust_amd import distributed as D
        common, size, jac = self.coracle.compare_matrix(self.sigs[lo:hi], self.sigs, num, 31, 0)
        cc = np.array([[len(np.intersect1d(a, b)) for b in self.sigs] for a in self.sigs[lo:hi]], dtype=np.int64).reshape(hi - lo, self.n)
        out = {"jaccard": jac.copy(), "common": common.view(np.int64).copy(), "size": size.view(np.int64).copy(), "count_common": cc}
        if ownership == 2 and hi > lo:
            i = np.arange(lo, hi)[:, None]
            j = np.arange(self.n)[None, :]
            final = D.owns(i, j, self.n) | ((j >= lo) & (j < hi))
            out["jaccard"][~final] = np.nan
            for k in ("common", "size", "count_common"):
                out[k][~final] = -7
        return {k: torch.from_numpy(out[k]) for k in want}

    def lengths(self, n_total):
        import torch
        return torch.tensor([len(s) for s in self.sigs], dtype=torch.int64)

    def close(self):
        pass


def _worker(rank, world, port, n_total, q, broken=None):
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

    if broken == "no_apply":          # the received blocks are dropped: the poisoned entries stay
        D.mirror_apply = lambda *a, **k: None
    elif broken == "shifted":         # every received block lands one row too low
        good = D.mirror_apply

        def shifted(out, recv, blocks, rank, n_total):
            good(out, [torch.roll(t.reshape(blocks[rank][1] - blocks[rank][0], -1), 1, 0).reshape(-1) if t.numel() else t for t in recv],
                 blocks, rank, n_total)
        D.mirror_apply = shifted
    want = ("jaccard", "common", "size", "count_common", "containment")
    out = D.compare_matrix_sharded(local_t, n_total, num, want=want, engine=OracleEngine(coracle))
    # the self-check bench.py runs at N > 1: sampled rows recomputed by this rank alone (ownership 0, its own dictionary)
    ver = D.verify_exchange(local_t, n_total, num, out, names=("jaccard", "common", "size", "count_common"), k_rows=4,
                            engine_factory=lambda: OracleEngine(coracle))
    if broken:
        q.put((rank, ver))
        dist.barrier()
        dist.destroy_process_group()
        return
    assert ver["ok"] and ver["rows_checked"] >= min(4, hi - lo), ver
    # and without the symmetric split: every pair of the row block computed locally, no all-to-all
    out_ns = D.compare_matrix_sharded(local_t, n_total, num, want=("jaccard",), engine=OracleEngine(coracle), symmetric=False)
    assert bool((out_ns["jaccard"] == out["jaccard"]).all())

    # the sketch side: every rank sketches its records, then the optional union
    recs_lo, recs_hi = D.shard_records(7, world, rank)
    mh = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
    for r in range(recs_lo, recs_hi):
        mh.add_sequence(bytes(coracle.synth_dna(r * 5000, 5000, 9, 0)), True)
    parts = D.merge_sketch_across_ranks(torch.from_numpy(mh.mins_np().view(np.int64)),
                                        torch.from_numpy(mh.abunds_np().view(np.int64)))
    q.put((rank, lo, hi, {k: v.numpy() for k, v in out.items()},
           [(m.numpy().view(np.uint64), a.numpy().view(np.uint64)) for m, a in parts]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 9), (2, 10), (4, 10), (4, 13), (4, 3)])
def test_row_sharded_matrix_and_sketch_union_gloo(world, n_total, coracle):
    """world-2 and world-4 runs of the sharded matrix over gloo: all-gather of the signatures, all-gather of the
    dictionary shares, pair ownership, the all-to-all of the mirrored blocks (the oracle engine poisons every entry a
    rank does not own, so only a complete exchange passes) -- against coracle.compare_matrix of the whole collection.
    (4, 13): short last block; (4, 3): an EMPTY last block; even and odd N (ties of the circular-half rule)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from sourmash_rust_amd import synth

    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    sigs = synth.family_signatures(0, n_total, num=200, n_families=3, pool=300, private=60, seed=5)
    rows = [sigs[i] for i in range(n_total)]
    common, size, jac = coracle.compare_matrix(rows, rows, 200, 31, 0)
    cc = np.array([[len(np.intersect1d(a, b)) for b in rows] for a in rows], dtype=np.int64)
    per = -(-n_total // world)
    assert [(r[1], r[2]) for r in res] == [(min(n_total, k * per), min(n_total, (k + 1) * per)) for k in range(world)]
    got = {k: np.concatenate([r[3][k] for r in res], axis=0) for k in res[0][3]}
    assert (got["jaccard"] == jac).all() and (got["common"].view(np.uint64) == common).all()
    assert (got["size"].view(np.uint64) == size).all() and (got["count_common"] == cc).all()
    assert (got["containment"] == cc.astype(np.float64) / 200.0).all()

    # union of the per-rank scaled sketches == sketch of all records on one rank (exact, abundances add)
    whole = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
    for r in range(7):
        whole.add_sequence(bytes(coracle.synth_dna(r * 5000, 5000, 9, 0)), True)
    for rank_res in res:
        merged = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
        for m, a in rank_res[4]:
            part = coracle.MinHash(0, 21, False, 42, 1 << 60, True)
            for h, c in zip(m, a):
                part.mins_push(int(h)); part.abunds_push(int(c))
            merged.merge(part)
        assert merged.mins == whole.mins and merged.abunds == whole.abunds


@pytest.mark.parametrize("broken", ["no_apply", "shifted"])
def test_a_broken_mirror_exchange_is_caught_by_verify_exchange(broken, coracle):
    """bench.py's N > 1 self-check (distributed.verify_exchange) must say NO, on every rank, when the exchange of the
    mirrored blocks is wrong: here mirror_apply is replaced by one that drops the received blocks, or shifts them by a row."""
    import torch.multiprocessing as mp
    world, n_total = 2, 12
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q, broken)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1]["ok"] is False for r in res), res          # the verdict is the conjunction over the ranks
    assert all(r[1]["rows_checked"] >= 4 for r in res)


def test_sample_row_stretches():
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from sourmash_rust_amd import distributed as D
    assert D.sample_row_stretches(5, 5, 16) == [] and D.sample_row_stretches(5, 9, 16) == [(5, 9)]
    st = D.sample_row_stretches(1250, 2500, 16)
    assert st[0][0] == 1250 and st[-1][1] == 2500 and sum(b - a for a, b in st) >= 16
    assert all(1250 <= a < b <= 2500 for a, b in st) and st == D.sample_row_stretches(1250, 2500, 16)


def test_pair_ownership_rule():
    """Every unordered pair has exactly one owner, every row owns floor(N/2) or so pairs, and block_needs() never
    says "nothing to send" for two blocks one of whose rows owns a pair with a row of the other."""
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    load_package()
    from sourmash_rust_amd import distributed as D
    for n in (1, 2, 3, 4, 7, 10, 11, 64):
        i = np.arange(n)[:, None]; j = np.arange(n)[None, :]
        own = D.owns(i, j, n)
        assert own.diagonal().all()
        off = ~np.eye(n, dtype=bool)
        assert ((own ^ own.T) | ~off).all(), n                 # exactly one of (i, j), (j, i)
        per_row = (own & off).sum(axis=1)
        assert per_row.max() - per_row.min() <= 1 and per_row.sum() == n * (n - 1) // 2
        for world in (1, 2, 3, 4, 8):
            blocks = [D.shard_range(n, world, r)[:2] for r in range(world)]
            for a, (alo, ahi) in enumerate(blocks):
                for b, (blo, bhi) in enumerate(blocks):
                    if a == b:
                        continue
                    truth = bool(own[alo:ahi, blo:bhi].any())
                    assert D.block_needs(alo, ahi, blo, bhi, n) or not truth, (n, world, a, b)


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (the shape of the driver's command, no torchrun around it) must
    start two ranks by itself.  The children see NO GPU (HIP/CUDA_VISIBLE_DEVICES are emptied before
    anything is launched, so this behaves the same on a CPU box, a 1-GPU box and a GPU node): each
    rank stops at its first line of GPU set-up -- which proves the launch went through: ranks ran
    with RANK/WORLD_SIZE set.  The real 2-rank run is tests/test_gpu_multirank.py."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    text = r.stdout + r.stderr
    # (the launcher stops the other rank as soon as one has failed, so only the first message is certain)
    assert "rank 0 needs cuda:0" in text or "rank 1 needs cuda:1" in text
    assert "but only 0 GPU(s) are visible" in text


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
