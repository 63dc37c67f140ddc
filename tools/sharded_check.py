"""Ranks of one job sharing ONE GPU (a rehearsal of the multi-GPU compare on a 1-GPU box): every rank computes its row
block of the all-vs-all matrix through distributed.compare_matrix_sharded over a real process group (gloo standing in
for RCCL, tensors staged through the host), rank 0 collects the blocks and checks them, bit for bit, against the matrix
one rank computes alone -- on the family, one-component and one-family collections.
    python -m torch.distributed.run --nproc-per-node W tools/sharded_check.py N"""
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
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    world, rank = dist.get_world_size(), dist.get_rank()
    ok = True
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
        for name in want:
            pad = torch.zeros((per, n), dtype=out[name].dtype)
            pad[: hi - lo] = out[name].cpu()
            parts = [torch.zeros_like(pad) for _ in range(world)] if rank == 0 else None
            dist.gather(pad, parts, dst=0)
            if rank == 0:
                got = torch.cat(parts)[:n]
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
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("sharded check", "ok" if ok else "FAILED", flush=True)
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
