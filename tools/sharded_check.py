"""The ranks of one job over a real process group: every rank computes its row block of the all-vs-all matrix through
distributed.compare_matrix_sharded, rank 0 collects the blocks and checks them, bit for bit, against the matrix one
rank computes alone -- on the family, one-component and one-family collections -- and every rank runs bench.py's
self-check (distributed.verify_exchange) on its own block.
Backend and device follow the machine: with at least as many GPUs as ranks, `nccl` (= RCCL over xGMI) and
cuda:LOCAL_RANK -- the real thing; with fewer (the 1-GPU development box), gloo standing in for RCCL (tensors staged
through the host) and every rank on cuda:0.  SHARDED_CHECK_BACKEND=gloo|nccl forces one.
A rank that finds a difference exits non-zero (and so does the launcher).
    python -m torch.distributed.run --nproc-per-node W tools/sharded_check.py N [GB per rank of the sketch-side check]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    from sourmash_rust_amd import distributed as D, synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    num = 300
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SHARDED_CHECK_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local if torch.cuda.device_count() >= world else 0)
        dist.init_process_group("gloo")
    assert (world, rank) == (dist.get_world_size(), dist.get_rank())
    if rank == 0:
        print("backend %s, %d rank(s), %d GPU(s) visible" % (backend, world, torch.cuda.device_count()), flush=True)
    ok, ok_mine = True, True
    for kind in ("families", "one_component", "one_family"):
        nf = 1 if kind == "one_family" else 9
        sigs = synth.family_signatures(0, n, num=num, n_families=nf, pool=2 * num, private=num // 2, seed=23)
        if kind == "one_component":
            sigs[:, 0] = 1
        lo, hi, per = D.shard_range(n, world, rank)
        blk = np.zeros((per, num), dtype=np.uint64)
        blk[: hi - lo] = sigs[lo:hi]
        mine = torch.from_numpy(blk.view(np.int64)).cuda()
        want = ("jaccard", "common", "count_common", "containment")
        out = D.compare_matrix_sharded(mine, n, num, want=want)
        ver = D.verify_exchange(mine, n, num, out, names=want, k_rows=16)
        ok_mine = ok_mine and ver["ok"]
        if rank == 0:
            print("%s verify_exchange world %d: %s (%d rows of rank 0)" % (kind, world, "equal" if ver["ok"] else "DIFFERENT", ver["rows_checked"]), flush=True)
        for name in want:
            pad = torch.zeros((per, n), dtype=out[name].dtype, device="cuda" if backend == "nccl" else "cpu")
            pad[: hi - lo] = out[name] if backend == "nccl" else out[name].cpu()
            # (an all-gather: nccl has no gather-to-one in every torch build; the matrix is small here)
            parts = [torch.zeros_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad)
            if rank == 0:
                got = torch.cat(parts)[:n].cpu()
                t = torch.from_numpy(sigs.view(np.int64)).cuda()
                off = np.arange(n + 1, dtype=np.uint64) * np.uint64(num)
                single = pkg.matrix.compare_block_dev(t, off, t, off, num, want=(name,))[name].cpu()
                same = bool((got == single).all())
                print("%s %s world %d: %s" % (kind, name, world, "equal" if same else "DIFFERENT"), flush=True)
                ok = ok and same
    # ---- the sketch side: every rank sketches its share of ONE input (protein arm, abundances), then the partial sketches
    # are united across ranks on the device (distributed.union_across_ranks); rank 0 compares with the sketch of everything
    gb = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    if gb > 0:
        import ctypes as C
        rlen, mx = 1_000_000, 18446744073709552
        nrec = int(gb * 1000)
        L = pkg.lib()

        def sketch(first, count):
            buf = torch.empty(count * rlen, dtype=torch.uint8, device="cuda")
            assert L.smh_synth_dna_dev(C.c_void_p(buf.data_ptr()), first * rlen, count * rlen, 5, 0, C.c_void_p(0)) == 0
            torch.cuda.synchronize()
            mh = pkg.KmerMinHash(0, 27, True, 42, mx, True)
            mh.add_sequences_dev(buf.data_ptr(), count * rlen, np.arange(count + 1, dtype=np.uint64) * np.uint64(rlen), True)
            return mh

        mine = sketch(rank * nrec, nrec)
        L.smh_profile_reset()
        uni = D.union_across_ranks(mine)
        ok_mine = ok_mine and D.verify_union(mine, uni)
        ms, k = C.c_double(), C.c_uint64()
        L.smh_profile_get(b"sketch_to_host", C.byref(ms), C.byref(k))
        none_copied = k.value == 0
        n_uni = len(uni)
        if rank == 0:
            whole = sketch(0, world * nrec)
            same = n_uni == len(whole) and uni.compare(whole) == 1.0 and (uni.mins_np() == whole.mins_np()).all() \
                and (uni.abunds_np() == whole.abunds_np()).all()
            print("union of %d ranks x %.1f GB (protein, abundances): %d hashes, %s, %s" %
                  (world, gb, n_uni, "equal" if same else "DIFFERENT", "nothing copied to the host" if none_copied else "COPIED TO HOST"), flush=True)
            ok = ok and same and none_copied
    flag = torch.tensor([1 if (ok and ok_mine) else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("sharded check", "ok" if flag.item() else "FAILED", flush=True)
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()
