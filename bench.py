#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Primary workload (configs[1]): sketch 10 GB of synthetic DNA per GPU (10 000 records x 1 MB,
k=31, scaled=1000 i.e. num=0, max_hash=18446744073709552, force=true), inputs already resident in
HBM when the timed region starts.  A step = one pass of the hot path over the batch: k-mer walk +
canonicalisation + murmur64 + filter (HIP), sort + distinct (HIP), merge into the sketch.
metric = k-mers hashed per second, whole job (all ranks).  Records are sharded across ranks
(weak scaling: 10 GB per GPU); there is no data-path collective in the sketch step.

Secondary (configs[2]/[3], reported in the same JSON line under "compare"): all-vs-all Jaccard
matrix of num=2000 signatures, rows sharded across ranks with one RCCL all-gather of the signatures.

Also on the line: "roofline" for the dominant kernel (k_dna_rolling) from HIP events recorded by the
library on the stream it launches on, and "cpu_baseline": the C oracle (a port of the reference's
algorithm; the reference is Rust and cannot be built here) timed on one host core over a bounded
sample of the same workload.

Launch: python bench.py --gpus 1 --steps K --warmup W
   or:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 31
MAX_HASH = 18446744073709552       # round((2^64-1)/1000): scaled=1000
REC_LEN = 1_000_000
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gb", type=float, default=10.0, help="GB of DNA per GPU (BASELINE config: 10)")
    ap.add_argument("--compare-n", type=int, default=0, help="signatures in the matrix (0 = by --gpus)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-compare", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus)
    # rehearsal knobs (not used by the driver): BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # BENCH_DIST_BACKEND=gloo swaps RCCL for gloo, so the N>1 code path can be exercised on a 1-GPU box
    if os.environ.get("BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from __graft_entry__ import load_package
    pkg = load_package()
    L = pkg.lib()
    stream = torch.cuda.current_stream().cuda_stream

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- input: resident in HBM
    n_rec = max(1, int(round(args.gb * 1e9 / REC_LEN)))
    total = n_rec * REC_LEN
    seq = torch.empty(total, dtype=torch.uint8, device="cuda")
    # every rank owns a different slice of one global synthetic stream (seed 2)
    rc = L.smh_synth_dna_dev(C.c_void_p(seq.data_ptr()), rank * total, total, 2, 0, C.c_void_p(stream))
    assert rc == 0, "synth failed"
    offsets = np.arange(n_rec + 1, dtype=np.uint64) * np.uint64(REC_LEN)
    kmers_per_step = n_rec * (REC_LEN - K + 1)

    def sketch_step():
        mh = pkg.KmerMinHash(0, K, False, 42, MAX_HASH, False)
        mh.add_sequences_dev(seq.data_ptr(), total, offsets, True, stream)
        return mh

    for _ in range(args.warmup):
        sketch_step()
    L.smh_profile_reset()
    L.smh_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mh = sketch_step()
    barrier()
    dt = time.perf_counter() - t0
    L.smh_profile_enable(0)
    retained = len(mh)          # the sketch is still in HBM here (DeviceSketch); bringing it to the
    t1 = time.perf_counter()    # host is a separate, untimed step whose cost is reported below
    host_mins = mh.mins_np()
    to_host_ms = (time.perf_counter() - t1) * 1e3
    assert host_mins.size == retained and bool((host_mins[1:] > host_mins[:-1]).all())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    value = world * kmers_per_step * args.steps / dt

    ms, launches = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"dna_rolling", C.byref(ms), C.byref(launches))
    kern_ms = ms.value / max(1, launches.value)
    launches_per_step = launches.value / max(1, args.steps)
    # algorithmic bytes per launch: 1 B read per k-mer position + 8 B per retained hash
    bytes_per_launch = (total + 8.0 * retained) / max(1.0, launches_per_step)
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, gfx950 x2
    # correction on FETCH_SIZE): measured on exactly this workload, kept under profiles/
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_dna_rolling.json")))
        if abs(total - 10e9) < 1 and launches_per_step == 1.0:
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pass
    # ceiling of any k=31 hashing kernel on this chip: bare murmur64 with operands in registers
    # (tools/microbench.hip, profiles/r01_microbench_int_ops.txt)
    MURMUR_CEILING = 361.4e9

    # ---------------------------------------------------------------- compare matrix (secondary)
    compare = None
    if not args.no_compare:
        from sourmash_rust_amd import distributed as D, synth
        n_sig = args.compare_n or {1: 1000, 2: 2500, 4: 5000, 8: 10000}.get(world, 1000 * world)
        lo, hi, per = D.shard_range(n_sig, world, rank)
        local_sigs = np.zeros((per, 2000), dtype=np.uint64)
        local_sigs[: hi - lo] = synth.family_signatures(lo, hi, num=2000, seed=3)
        mine = torch.from_numpy(local_sigs.view(np.int64)).cuda()

        def compare_step():
            # rows sharded by contiguous blocks; ONE all-gather (RCCL over xGMI) of the signatures
            return D.compare_matrix_sharded(mine, n_sig, 2000, want=("jaccard",))

        out = compare_step()
        barrier()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            out = compare_step()
        barrier()
        cdt = (time.perf_counter() - t0) / reps
        ct = torch.tensor([cdt], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(ct, op=dist.ReduceOp.MAX)
        cdt = float(ct.item())
        diag_ok = True
        if hi > lo:
            j = out["jaccard"]
            idx = torch.arange(hi - lo, device="cuda")
            diag_ok = bool((j[idx, idx + lo] == 1.0).all().item())
        # every rank checks its own row block; the line reports the conjunction
        okt = torch.tensor([1 if diag_ok else 0], dtype=torch.int64, device="cuda")
        if world > 1:
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        diag_ok = bool(okt.item())
        tv, tt, ppt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.smh_compare_last_stats(C.byref(tv), C.byref(tt), C.byref(ppt))
        visited = min(tv.value * ppt.value, (hi - lo) * n_sig)          # pairs actually walked on this rank
        eff = (visited * 32008 + ((hi - lo) * n_sig - visited) * 8) / cdt / 1e9
        compare = {"metric": "signature pairs compared/sec (ordered pairs, num=2000)", "value": n_sig * n_sig / cdt,
                   "unit": "pairs/s", "n_signatures": n_sig, "seconds": cdt, "self_jaccard_is_1": diag_ok,
                   "tiles_visited": tv.value, "tiles_total": tt.value, "pairs_per_tile": ppt.value,
                   "collection": "50 families of related signatures (SURVEY.md 8d); pairs across families share no hash "
                                 "and are filled without being read (DESIGN.md 3.4)",
                   "roofline": {"bound": "hbm", "achieved": eff, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": eff / HBM_PEAK_GBS,
                                "label": "rank 0's EFFECTIVE bytes: 32 008 B per ordered pair walked (SURVEY.md 8d) + 8 B per pair "
                                         "filled as disjoint; tiles are served from LDS/L2, compulsory HBM traffic is N*16 KB in + "
                                         "N^2*8 B out; the kernel is bound by VALU issue and LDS latency (DESIGN.md 3.4)"}}

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import coracle
        o = coracle.MinHash(0, K, False, 42, MAX_HASH, False)
        done = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.cpu_seconds and done < n_rec:
            o.add_sequence(bytes(coracle.synth_dna(done * REC_LEN, REC_LEN, 2, 0)), True)
            done += 1
        cdt = time.perf_counter() - t0
        # the same records through the GPU path must give the same sketch
        g = pkg.KmerMinHash(0, K, False, 42, MAX_HASH, False)
        g.add_sequences_dev(seq.data_ptr(), done * REC_LEN, offsets[: done + 1], True, stream)
        assert g.mins == o.mins, "GPU sketch differs from the CPU oracle on the baseline sample"
        cpu = {"value": done * (REC_LEN - K + 1) / cdt, "unit": "k-mers/s", "cores": 1, "kind": "port",
               "sample": "first %d of the %d records (1 MB each) of the same workload, C oracle incl. input generation; "
                         "sketch checked equal to the GPU's" % (done, n_rec)}

    if rank == 0:
        line = {
            "metric": "k-mers hashed/sec (k=31, scaled=1000)", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "sketch %.1f GB synthetic DNA per GPU, %d records x 1 MB, k=31, num=0, "
                                   "max_hash=%d (scaled=1000), force=true, inputs resident in HBM" % (total / 1e9, n_rec, MAX_HASH),
                       "retained_hashes": retained, "records_sharded_across_ranks": True,
                       "result": "sorted distinct hashes left in HBM; copy to host (PCIe, not in value) took %.1f ms" % to_host_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_dna_rolling<31,512,2>",
                         "kernel_ms_avg": kern_ms, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu_bound": {"kernel_kmers_per_s": kmers_per_step / max(1.0, launches_per_step) / (kern_ms * 1e-3) if kern_ms > 0 else 0.0,
                                        "bare_murmur64_ceiling_per_s": MURMUR_CEILING,
                                        "frac": (kmers_per_step / max(1.0, launches_per_step) / (kern_ms * 1e-3) / MURMUR_CEILING) if kern_ms > 0 else 0.0},
                         "note": "1 B/k-mer: the path is integer-VALU bound (34 multiply-class + ~105 other VALU ops per k-mer; a third of murmur's "
                                 "multiplies come from LDS product tables, which is why the kernel can approach the straightforward bare-murmur rate), "
                                 "not HBM bound; see DESIGN.md 'Roofline'"},
            "cpu_baseline": cpu,
            "compare": compare,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
