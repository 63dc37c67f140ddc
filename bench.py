#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X.

Primary workload (configs[1]): sketch 10 GB of synthetic DNA per GPU (10 000 records x 1 MB,
k=31, scaled=1000 i.e. num=0, max_hash=18446744073709552, force=true), inputs already resident in
HBM when the timed region starts.  A step = one pass of the hot path over the batch: k-mer walk +
canonicalisation + murmur64 + filter (HIP), sort + distinct (HIP), merge into the sketch.
metric = k-mers hashed per second, whole job (all ranks).  Records are sharded across ranks
(weak scaling: 10 GB per GPU); there is no data-path collective in the sketch step.

Secondary (configs[2]/[3], reported in the same JSON line under "compare"): all-vs-all Jaccard
matrix of num=2000 signatures.  N = 10 000 at EVERY world size (strong scaling: the 10 000 x 10 000
matrix of configs[3] on 1, 2, 4, 8 GPUs; the 1 000 x 1 000 block of configs[2] additionally at
world 1), row blocks sharded across ranks: all-gather of the signatures, the dictionary pre-pass
sharded by hash range + one all-gather of its shares, every rank walks the pairs its rows OWN
(half of its row block: the walk is symmetric) and one all-to-all hands over the mirrored blocks
(sourmash-rust_amd/distributed.py) -- on the family-structured collection of SURVEY.md 8d, on the
same collection with one hash shared by every signature (a contaminant k-mer), and on ONE family
(every pair has to be walked), each next to the C oracle on the host cores.

Also on the line: "roofline" for the dominant kernel (k_dna_rolling) from HIP events recorded by the
library on the stream it launches on, and "cpu_baseline": the C oracle (a port of the reference's
algorithm; the reference is Rust and cannot be built here) timed on ONE host core and on all of
them (count printed) over bounded samples of the same workloads, results checked equal to the GPU's.

Launch: python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own N ranks)
   or:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import ctypes as C
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 31
MAX_HASH = 18446744073709552       # round((2^64-1)/1000): scaled=1000
REC_LEN = 1_000_000
NUM = 2000                         # signature size of the compare matrix (configs[2], [3])
PROT_K = 27                        # configs[4]: protein arm, ksize=27 = windows of 9 residues
PROT_SEED = 5                      # generator seed of the C5 input
PROT_WINDOWS_PER_REC = sum(2 * max(0, (REC_LEN - f) // 3 - PROT_K // 3 + 1) for f in range(3))   # six frames
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMDS = 256 * 4                    # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32
VALU_CYCLES = 2.0                  # ... a wave64 VALU instruction issues over 2 cycles
MAX_CLOCK_HZ = 2.4e9               # ... max clock
LDS_READ_B32_PEAK_TBS = 75.0       # ... LDS aggregate for ds_read_b32, every CU streaming
CONTAMINANT = 1                    # the hash every signature of the one-component collection shares


def host_cores():
    """Cores this process may use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return n


# --------------------------------------------------------------------------------------------
# CPU-baseline workers: separate processes (started as children; they never touch the GPU) that
# run the C oracle over their share of the sample and leave the result where the parent can check
# it against the GPU's.  `python bench.py --cpu-worker <kind> ...`
def cpu_worker(argv):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import coracle
    kind, out_path = argv[0], argv[1]
    if kind == "sketch":
        first, last, budget = int(argv[2]), int(argv[3]), float(argv[4])
        o = coracle.MinHash(0, K, False, 42, MAX_HASH, False)
        done, spent = 0, 0.0
        for r in range(first, last):
            rec = bytes(coracle.synth_dna(r * REC_LEN, REC_LEN, 2, 0))      # input generation is NOT timed
            t0 = time.perf_counter()
            o.add_sequence(rec, True)
            spent += time.perf_counter() - t0
            done += 1
            if spent >= budget:
                break
        np.save(out_path, o.mins_np())
        print(json.dumps({"records": done, "seconds": spent}))
    elif kind == "protein":
        # BASELINE configs[4], a sample of one rank's share: six-frame translation, ksize=27 (9 residues), scaled, abundances
        first, last, budget = int(argv[2]), int(argv[3]), float(argv[4])
        o = coracle.MinHash(0, PROT_K, True, 42, MAX_HASH, True)
        done, spent = 0, 0.0
        for r in range(first, last):
            rec = bytes(coracle.synth_dna(r * REC_LEN, REC_LEN, PROT_SEED, 0))
            t0 = time.perf_counter()
            o.add_sequence(rec, True)
            spent += time.perf_counter() - t0
            done += 1
            if spent >= budget:
                break
        np.save(out_path, np.stack([o.mins_np(), o.abunds_np()]))
        print(json.dumps({"records": done, "seconds": spent}))
    elif kind == "config0":
        # BASELINE configs[0]: 1 MB, k=31, num=500, compare to itself (reference tests/minhash.rs path)
        rec = bytes(coracle.synth_dna(0, REC_LEN, 1, 0))
        o = coracle.MinHash(500, K, False, 42, 0, False)
        t0 = time.perf_counter()
        o.add_sequence(rec, False)
        t1 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            j = o.compare(o)
        t2 = time.perf_counter()
        np.save(out_path, o.mins_np())
        print(json.dumps({"sketch_seconds": t1 - t0, "kmers": REC_LEN - K + 1, "compare_seconds": (t2 - t1) / reps,
                          "self_compare": j}))
    elif kind == "compare":
        n_sig, first, last, budget, kind_id = int(argv[2]), int(argv[3]), int(argv[4]), float(argv[5]), int(argv[6])
        if len(argv) > 7:
            sigs = np.load(argv[7], mmap_mode="r")          # the parent's copy of the collection (generation not timed)
        else:
            from __graft_entry__ import load_package
            load_package()
            sigs = collection(kind_id, 0, n_sig)
        cols = [sigs[i] for i in range(n_sig)]
        rows_done, spent, jac = 0, 0.0, []
        for r in range(first, last):
            t0 = time.perf_counter()
            _, _, j = coracle.compare_matrix([sigs[r]], cols, NUM, K, 0)
            spent += time.perf_counter() - t0
            jac.append(j[0])
            rows_done += 1
            if spent >= budget:
                break
        np.save(out_path, np.stack(jac) if jac else np.zeros((0, n_sig)))
        print(json.dumps({"rows": rows_done, "seconds": spent}))
    else:
        raise SystemExit("unknown worker kind " + kind)


def collection(kind_id, lo, hi):
    """Signatures lo..hi-1 of the compare collections: 0 = 50 families; 1 = the same with one hash
    shared by every signature; 2 = ONE family (every pair shares hundreds of hashes)."""
    from sourmash_rust_amd import synth
    if kind_id == 2:
        return synth.family_signatures(lo, hi, num=NUM, n_families=1, seed=3)
    sigs = synth.family_signatures(lo, hi, num=NUM, seed=3)
    if kind_id == 1:
        sigs[:, 0] = CONTAMINANT       # (uniform 64-bit hashes: the smallest possible value keeps rows ascending)
    return sigs


def run_workers(specs):
    """specs: list of argv lists.  Starts them all at once, returns [(parsed json, output path)]."""
    tmp = tempfile.mkdtemp(prefix="smh_cpu_")
    TMP_DIRS.append(tmp)
    procs = []
    for i, spec in enumerate(specs):
        out = os.path.join(tmp, "w%d.npy" % i)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-worker", spec[0], out] + [str(x) for x in spec[1:]]
        procs.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True), out))
    res = []
    try:
        for p, out in procs:
            stdout, _ = p.communicate(timeout=600)
            if p.returncode != 0:
                raise RuntimeError("cpu worker failed: " + stdout[-500:])
            res.append((json.loads(stdout.strip().splitlines()[-1]), out))
    finally:
        # a worker that timed out or failed must not leave the others burning the host's cores
        for p, _ in procs:
            if p.poll() is None:
                p.kill()
        for p, _ in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
    return res


TMP_DIRS = []          # scratch directories of the CPU-baseline workers; removed when the line has been printed


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks as fresh child
    processes (this process has not touched the GPU), relay their output, exit with their code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + argv
    return subprocess.call(cmd, cwd=ROOT)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gb", type=float, default=10.0, help="GB of DNA per GPU (BASELINE config: 10)")
    ap.add_argument("--compare-n", type=int, default=0, help="signatures in the matrix (0 = by --gpus)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of the 1-core sketch sample (others scale with it)")
    ap.add_argument("--no-compare", action="store_true")
    ap.add_argument("--protein-gb", type=float, default=12.5, help="GB of DNA per GPU through the protein arm (configs[4]: 100 GB / 8 GPUs; 0 = skip)")
    ap.add_argument("--host-gb", type=float, default=2.0, help="GB of the workload also timed from HOST memory (PCIe-inclusive; 0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))

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
    if local >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d needs cuda:%d but only %d GPU(s) are visible" % (rank, local, torch.cuda.device_count()))
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

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

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
    # N > 1: the ranks' partial sketches united into ONE sketch on the device (one all-gather of the padded hash arrays +
    # rank-arithmetic unions, distributed.union_across_ranks) -- reported on its own, never part of `value`
    union = None
    if world > 1:
        from sourmash_rust_amd import distributed as D
        barrier()
        t0 = time.perf_counter()
        uni = D.union_across_ranks(mh)
        barrier()
        union_s = max_over_ranks(time.perf_counter() - t0)
        vt = torch.tensor([1 if D.verify_union(mh, uni) else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(vt, op=dist.ReduceOp.MIN)
        union = {"union_ms": union_s * 1e3, "hashes": len(uni), "parts": world, "verified": bool(vt.item()),
                 "verified_what": "on every rank, in HBM: the union holds each of the rank's own hashes, with a count >= the rank's own",
                 "what": "the %d ranks' partial sketches -> one sketch on every rank, in HBM (all-gather of the hash arrays over RCCL, "
                         "then per part a union by rank arithmetic and two scatters; KmerMinHash::merge semantics, "
                         "reference src/lib.rs:307-403).  Not part of `value`" % world}
        del uni
    retained = len(mh)          # the sketch is still in HBM here (DeviceSketch); bringing it to the
    t1 = time.perf_counter()    # host is a separate, untimed step whose cost is reported below
    host_mins = mh.mins_np()
    to_host_ms = (time.perf_counter() - t1) * 1e3
    assert host_mins.size == retained and bool((host_mins[1:] > host_mins[:-1]).all())

    dt = max_over_ranks(dt)
    value = world * kmers_per_step * args.steps / dt

    ms, launches = C.c_double(), C.c_uint64()
    L.smh_profile_get(b"dna_rolling", C.byref(ms), C.byref(launches))
    kern_ms = ms.value / max(1, launches.value)
    launches_per_step = launches.value / max(1, args.steps)
    # algorithmic bytes per launch: 1 B read per k-mer position + 8 B per retained hash
    bytes_per_launch = (total + 8.0 * retained) / max(1.0, launches_per_step)
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    kmers_per_launch = kmers_per_step / max(1.0, launches_per_step)
    # Counter-derived figures come from the committed rocprofv3 --pmc passes of exactly this workload
    # (profiles/), NOT from this run: HBM bytes per launch (FETCH_SIZE x2 on gfx950 + WRITE_SIZE) and
    # VALU wave-instructions per 64 k-mers (SQ_INSTS_VALU / wave steps).
    traffic, traffic_src, valu, half_share = None, None, None, None
    for name in ("r04_pmc_dna_rolling.json", "r03_pmc_dna_rolling.json", "r02_pmc_dna_rolling.json", "r01_pmc_dna_rolling.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if abs(total - 10e9) < 1 and launches_per_step == 1.0:
            traffic = pmc.get("hbm_bytes_per_launch")
            valu = pmc.get("valu_insts_per_64_kmers")
            if pmc.get("static_half_rate_per_kmer") is not None:
                half_share = pmc["static_half_rate_per_kmer"] / (pmc["static_half_rate_per_kmer"] + pmc["static_full_rate_per_kmer"])
            traffic_src = "profiles/%s: separate rocprofv3 --pmc passes of this command, committed; not measured in this run" % name
        break
    valu_bound = None
    if valu and kern_ms > 0:
        floor_ms = kmers_per_launch / 64.0 * valu * VALU_CYCLES / (SIMDS * MAX_CLOCK_HZ) * 1e3
        valu_bound = {"valu_insts_per_64_kmers": valu, "cycles_per_wave_inst": VALU_CYCLES, "simds": SIMDS,
                      "clock_hz": MAX_CLOCK_HZ, "floor_ms": floor_ms, "frac": floor_ms / kern_ms,
                      "kernel_kmers_per_s": kmers_per_launch / (kern_ms * 1e-3),
                      "label": "VALU issue floor from MI355X_MICROARCH.md (2 cycles per wave64 instruction, 1024 SIMDs, 2.4 GHz) "
                               "for the kernel's measured instruction count (SQ_INSTS_VALU, committed PMC pass); "
                               "frac = floor / measured kernel time.  NB: relative to the kernel's OWN instruction count -- "
                               "removing instructions lowers it; kernel_kmers_per_s is the absolute figure"}
        if half_share is not None:
            # the share of half-rate instructions (multiplies, v_mad_u64_u32, three-operand forms: 4 cycles each) from the
            # static count of the loop (tools/isa_count.py), applied to the measured instruction count
            wfloor = floor_ms * (1.0 + half_share)
            valu_bound["rate_weighted"] = {"half_rate_share": half_share, "floor_ms": wfloor, "frac": wfloor / kern_ms,
                                           "label": "the same floor with the kernel's half-rate instructions priced at 4 cycles"}
            mc = pmc.get("measured_cycles_per_wave_inst")
            if mc:
                # ... and at the issue rates the microbenchmark measured on this chip for the two classes
                mfloor = floor_ms / VALU_CYCLES * ((1.0 - half_share) * mc["two_operand"] + half_share * mc["slow_class"])
                valu_bound["at_measured_rates"] = {"cycles_two_operand": mc["two_operand"], "cycles_slow_class": mc["slow_class"],
                                                   "floor_ms": mfloor, "frac": mfloor / kern_ms,
                                                   "label": "VALU issue time of the kernel's instruction mix at the measured issue rates "
                                                            "(profiles/r02_instr_rates.txt); what is left is attributed in DESIGN.md section 7"}

    # ---------------------------------------------------------------- protein arm (configs[4], one rank's share)
    protein, pseq, poffsets = None, None, None
    if args.protein_gb > 0:
        pn_rec = max(1, int(round(args.protein_gb * 1e9 / REC_LEN)))
        ptotal = pn_rec * REC_LEN
        pseq = torch.empty(ptotal, dtype=torch.uint8, device="cuda")
        assert L.smh_synth_dna_dev(C.c_void_p(pseq.data_ptr()), rank * ptotal, ptotal, PROT_SEED, 0, C.c_void_p(stream)) == 0
        poffsets = np.arange(pn_rec + 1, dtype=np.uint64) * np.uint64(REC_LEN)

        def protein_step():
            m = pkg.KmerMinHash(0, PROT_K, True, 42, MAX_HASH, True)
            m.add_sequences_dev(pseq.data_ptr(), ptotal, poffsets, True, stream)
            return m

        for _ in range(max(1, args.warmup)):
            protein_step()
        L.smh_profile_reset()
        L.smh_profile_enable(1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pmh = protein_step()
        barrier()
        pdt = max_over_ranks(time.perf_counter() - t0)
        L.smh_profile_enable(0)
        pms, plaunch = C.c_double(), C.c_uint64()
        L.smh_profile_get(b"protein_fused", C.byref(pms), C.byref(plaunch))
        pk_ms = pms.value / max(1, plaunch.value)
        p_retained = len(pmh)
        windows_per_step = pn_rec * PROT_WINDOWS_PER_REC
        # algorithmic bytes: 1 B read per base = 0.5 B per window, + 8 B per candidate hash written
        p_bytes = ptotal + 8.0 * p_retained
        p_ach = p_bytes / (pk_ms * 1e-3) / 1e9 if pk_ms > 0 else 0.0
        ppmc = None
        for name in ("r04_pmc_protein_fused.json", "r03_pmc_protein_fused.json"):
            try:
                ppmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                ppmc["_file"] = name
                break
            except Exception:
                continue
        p_valu = None
        if ppmc and pk_ms > 0 and abs(ptotal - 12.5e9) < 1 and plaunch.value == args.steps:
            vi = ppmc["valu_insts_per_64_positions"]
            floor = ptotal / 64.0 * vi * VALU_CYCLES / (SIMDS * MAX_CLOCK_HZ) * 1e3
            p_valu = {"valu_insts_per_64_positions": vi, "floor_ms": floor, "frac": floor / pk_ms,
                      "label": "VALU issue floor (2 cycles per wave64 instruction, 1024 SIMDs, 2.4 GHz) for the kernel's measured "
                               "instruction count: two murmur hashes per position (forward and reverse-complement window)",
                      "source": "profiles/%s (separate rocprofv3 --pmc passes, committed; not measured in this run)" % ppmc["_file"]}
        p_union = None
        if world > 1:
            from sourmash_rust_amd import distributed as D
            barrier()
            t0 = time.perf_counter()
            puni = D.union_across_ranks(pmh)
            barrier()
            pu_s = max_over_ranks(time.perf_counter() - t0)
            vt = torch.tensor([1 if D.verify_union(pmh, puni) else 0], dtype=torch.int64, device="cuda")
            dist.all_reduce(vt, op=dist.ReduceOp.MIN)
            p_union = {"union_ms": pu_s * 1e3, "hashes": len(puni), "parts": world, "verified": bool(vt.item()),
                       "what": "configs[4]'s last step: the %d ranks' partial sketches (hashes + abundances) -> ONE signature on every "
                               "rank, in HBM (distributed.union_across_ranks).  Not part of `value`" % world}
            del puni
        protein = {"metric": "protein windows hashed/sec (six-frame translation, ksize=%d, scaled=1000, abundance)" % PROT_K,
                   "value": world * windows_per_step * args.steps / pdt, "unit": "windows/s", "ms_per_step": pdt / args.steps * 1e3,
                   "bases_per_s": world * ptotal * args.steps / pdt, "scaling": "weak",
                   "config": {"workload": "BASELINE configs[4], one rank's share of the 100 GB: %.1f GB synthetic DNA per GPU, %d records x 1 MB, "
                                          "is_protein, ksize=%d (windows of %d residues, 6 frames), num=0, max_hash=%d, track_abundance, "
                                          "force=true, inputs resident in HBM" % (ptotal / 1e9, pn_rec, PROT_K, PROT_K // 3, MAX_HASH),
                              "retained_hashes": p_retained, "windows_per_step": windows_per_step},
                   "roofline": {"bound": "hbm", "achieved": p_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": p_ach / HBM_PEAK_GBS,
                                "traffic": ppmc.get("hbm_bytes_per_launch") if p_valu else None,
                                "kernel": "k_protein_fused<%d,512>" % (PROT_K // 3), "kernel_ms_avg": pk_ms,
                                "launches_per_step": plaunch.value / max(1, args.steps), "algorithmic_bytes_per_launch": p_bytes,
                                "valu_bound": p_valu,
                                "note": "0.5 B per window: like the DNA arm, integer-VALU bound, not HBM bound (reference src/lib.rs:275-302)"},
                   "union_across_ranks": p_union, "cpu_baseline": None}

    # ---------------------------------------------------------------- compare matrix (secondary)
    compare = None
    cmp_outs = {}
    if not args.no_compare:
        from sourmash_rust_amd import distributed as D, matrix as MX
        # strong scaling: N fixed at 10 000 (configs[3]) whatever the world size; world 1 also runs configs[2]'s 1 000
        sizes = [args.compare_n] if args.compare_n else ([10000, 1000] if world == 1 else [10000])

        def prof(name):
            ms, k = C.c_double(), C.c_uint64()
            L.smh_profile_get(name, C.byref(ms), C.byref(k))
            return ms.value, k.value

        def time_collection(n_sig, kind_id):
            lo, hi, per = D.shard_range(n_sig, world, rank)
            blk = np.zeros((per, NUM), dtype=np.uint64)
            blk[: hi - lo] = collection(kind_id, lo, hi)
            mine = torch.from_numpy(blk.view(np.int64)).cuda()

            def compare_step(timings=None):
                return D.compare_matrix_sharded(mine, n_sig, NUM, want=("jaccard",), timings=timings)

            out = compare_step()
            L.smh_profile_reset()
            L.smh_profile_enable(1)
            barrier()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                out = compare_step()
            barrier()
            cdt = max_over_ranks((time.perf_counter() - t0) / reps)
            L.smh_profile_enable(0)
            kern = {k: prof(k.encode()) for k in ("compare_tiled", "compare_comp", "compare_fill")}
            phases = {}
            compare_step(phases)             # one more, untimed pass with a synchronisation after every phase
            diag_ok = True
            if hi > lo:
                j = out["jaccard"]
                idx = torch.arange(hi - lo, device="cuda")
                diag_ok = bool((j[idx, idx + lo] == 1.0).all().item())
            # every rank checks its own row block; the line reports the conjunction
            okt = torch.tensor([1 if diag_ok else 0], dtype=torch.int64, device="cuda")
            if world > 1:
                dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            # ... and, untimed: every output once more through the sharded path, then ~16 sampled rows of every rank's block
            # recomputed by that rank ALONE (its own world-1 dictionary, ownership 0: every pair of the row walked locally,
            # nothing mirrored or received) and compared bit for bit -- so that the first run over real RCCL says by itself
            # whether the sliced dictionary, the ownership rule and the exchange of the mirrored blocks were right
            st = MX.last_stats()             # (of the timed configuration: the check below runs small blocks of its own)
            names = ("jaccard", "common", "size", "count_common", "containment")
            full = D.compare_matrix_sharded(mine, n_sig, NUM, want=names)
            same_again = bool((full["jaccard"] == out["jaccard"]).all().item()) if hi > lo else True
            ver = D.verify_exchange(mine, n_sig, NUM, full, names=names, k_rows=16)
            del full
            walked = min(st["tiles_visited"] * st["pairs_per_tile"], (hi - lo) * n_sig)   # pairs walked on this rank
            kname = "compare_tiled" if st["route"] == "tiled" else "compare_comp"
            kms = kern[kname][0] / max(1, kern[kname][1])
            rec = {"seconds": cdt, "pairs_per_s": n_sig * n_sig / cdt, "self_jaccard_is_1": bool(okt.item()),
                   "exchange_verified": bool(ver["ok"] and same_again), "verified_rows_rank0": ver["rows_checked"],
                   "route": st["route"], "tiles_visited": st["tiles_visited"], "tiles_total": st["tiles_total"],
                   "pairs_per_tile": st["pairs_per_tile"], "rank0_pairs_walked": walked,
                   "rank0_kernel": "k_compare_" + (("tiled_pf" if st.get("pipelined") else "tiled") if st["route"] == "tiled" else "comp"), "rank0_kernel_ms": kms,
                   "rank0_fill_ms": kern["compare_fill"][0] / max(1, kern["compare_fill"][1]),
                   "rank0_pairs_walked_per_s": walked / (kms * 1e-3) if kms > 0 else None,
                   "rank0_union_elements_walked_per_s": walked * NUM / (kms * 1e-3) if kms > 0 else None,
                   "rank0_phase_ms": {k: v * 1e3 for k, v in phases.items()}}
            return out, rec

        per_size = {}
        for n_sig in sizes:
            per_size[n_sig] = {}
            for key, kind_id in (("families", 0), ("one_component", 1), ("one_family", 2)):
                out, rec = time_collection(n_sig, kind_id)
                per_size[n_sig][key] = rec
                if world == 1 and rank == 0 and args.cpu_seconds > 0:
                    # rows of the first 1 000-row block, on the host: what the CPU baseline's rows are compared with
                    cmp_outs.setdefault(n_sig, {})[key] = out["jaccard"][: min(n_sig, 1000)].cpu().numpy()
                del out
        head = per_size[sizes[0]]
        dense = head["one_family"]
        # roofline of the matrix kernel on the collection where every pair is walked (rank 0's kernel, HIP events)
        cmp_roof = None
        if dense["rank0_kernel_ms"] > 0 and dense["route"] == "tiled":
            pmc, pmc_file = None, None
            # (range masks since round 4, at every world size: counters of that kernel)
            for name in ("r04_pmc_compare_tiled.json", "r03_pmc_compare_tiled.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                    pmc_file = name
                    break
                except Exception:
                    continue
            walked, kms = dense["rank0_pairs_walked"], dense["rank0_kernel_ms"]
            masked = bool(pmc and pmc.get("bound") == "issue")
            lds_bytes = pmc.get("lds_bytes_per_walked_pair") * walked if pmc and pmc.get("lds_bytes_per_walked_pair") else None
            lds_achieved = lds_bytes / (kms * 1e-3) / 1e12 if lds_bytes else None
            cmp_roof = {"kernel": dense["rank0_kernel"] + (" (range masks)" if masked else ""), "kernel_ms_avg": kms, "pairs_walked": walked,
                        "n_signatures": sizes[0],
                        "effective_bytes": walked * ((NUM + NUM) * 8 + 8),
                        "effective_TBps": walked * ((NUM + NUM) * 8 + 8) / (kms * 1e-3) / 1e12,
                        "compulsory_hbm_bytes": sizes[0] * NUM * 8 + D.shard_range(sizes[0], world, 0)[2] * sizes[0] * 8,
                        "counters_source": "profiles/%s (separate rocprofv3 --pmc passes of %s on the world-1 block, committed; not measured in this run)"
                                           % (pmc_file, pmc.get("kernel", "the kernel")) if pmc else None,
                        "valu_bound": None}
            if masked:
                # range masks (round 4; at world > 1 built from the flags the slice owners send): per pair the range of its cut comes
                # from popcounts and only that range is walked -- next to no LDS traffic, few loads; what bounds the kernel is
                # instruction issue, vector and scalar
                vi, si = pmc["valu_wave_insts_per_64_pairs"], pmc.get("salu_wave_insts_per_64_pairs")
                valu_rate = walked / 64.0 * vi / (kms * 1e-3) / 1e9                  # G wave-instructions per second, this run's time
                valu_peak = SIMDS * MAX_CLOCK_HZ / VALU_CYCLES / 1e9
                cmp_roof.update({"bound": "instruction issue (VALU + scalar)", "unit": "G VALU wave-instructions/s", "peak": valu_peak,
                                 "achieved": valu_rate, "frac": valu_rate / valu_peak,
                                 "scalar_unit_busy": (walked / 64.0 * si / 256.0) / (kms * 1e-3 * MAX_CLOCK_HZ) if si else None,
                                 "wait_any_share_of_wave_cycles": pmc.get("wait_any_share_of_wave_cycles"),
                                 "vmem_loads_per_64_pairs": pmc.get("vmem_loads_per_64_pairs"),
                                 "memory_side_bytes": pmc.get("memory_side_bytes_per_launch"),
                                 "note": "integer compare/indexing, no MFMA.  The union of a pair is cut after `num` elements (reference "
                                         "src/lib.rs:470-499); where that happens follows from per-range bit masks of the shared hashes "
                                         "(DESIGN.md 3.4 'range masks'), so a pair costs ~1 900 VALU + ~1 500 scalar wave-instructions per 64 "
                                         "pairs instead of ~27 000 VALU.  `peak` = the guide's VALU issue rate (2 cycles per wave64 "
                                         "instruction, 1024 SIMDs, 2.4 GHz); most of the kernel's VALU instructions are in the slow class "
                                         "(~4.3 cycles measured), and the one scalar unit of a CU is busy `scalar_unit_busy` of the time.  "
                                         "'effective' = SURVEY.md 8d's (|A|+|B|)*8+8 B per compared pair over the kernel time (not traffic: "
                                         "most of those bytes are never read)"})
            else:
                cmp_roof.update({"bound": "lds", "unit": "TB/s", "peak": LDS_READ_B32_PEAK_TBS,
                                 "achieved": lds_achieved, "frac": lds_achieved / LDS_READ_B32_PEAK_TBS if lds_achieved else None,
                                 "lds_bytes": lds_bytes,
                                 "note": "integer compare/indexing, no MFMA.  'effective' = SURVEY.md 8d's (|A|+|B|)*8+8 B per walked pair, "
                                         "served from LDS/L2 (may exceed the HBM peak: it is not HBM traffic); 'achieved' = actual ds_read "
                                         "bytes against the guide's ds_read_b32 aggregate (~75 TB/s); compulsory HBM traffic is "
                                         "N*16 KB in + rows*N*8 B out"})
        if cmp_roof and pmc and pmc.get("valu_wave_insts_per_64_pairs"):
            # VALU issue: wave instructions per 64 compared pairs from the committed PMC pass at the guide's 2 cycles per wave64 instruction
            vi = pmc["valu_wave_insts_per_64_pairs"]
            floor2 = cmp_roof["pairs_walked"] / 64.0 * vi * VALU_CYCLES / (SIMDS * MAX_CLOCK_HZ) * 1e3
            cmp_roof["valu_bound"] = {"valu_wave_insts_per_64_pairs": vi, "floor_ms_at_2_cycles": floor2,
                                      "frac_at_2_cycles": floor2 / cmp_roof["kernel_ms_avg"],
                                      "per_merge_step": pmc.get("per_merge_step")}
        compare = {"metric": "signature pairs compared/sec (ordered pairs delivered, num=%d)" % NUM, "value": head["families"]["pairs_per_s"],
                   "unit": "pairs/s", "n_signatures": sizes[0], "seconds": head["families"]["seconds"],
                   "value_every_pair_walked": head["one_family"]["pairs_per_s"],
                   "value_note": "`value` is the family collection of SURVEY.md 8d, where 98 % of the pairs share nothing and are filled "
                                 "without being walked; `value_every_pair_walked` is the ONE-family collection (every pair walked): the "
                                 "kernel's own rate, and the figure to compare kernels by",
                   "scaling": "strong: N = %d signatures at every world size (the %d x %d matrix of BASELINE configs[3])" % (sizes[0], sizes[0], sizes[0]),
                   "symmetry": "used: all signatures have one num, so the union walk is symmetric; row i owns the pairs (i, j) with "
                               "(j - i) mod N < N/2, every rank walks the pairs its rows own and the mirrored blocks are exchanged "
                               "(world 1: upper triangle + mirror writes).  pairs/s counts ORDERED pairs delivered (N^2)",
                   "self_jaccard_is_1": all(v["self_jaccard_is_1"] for v in head.values()),
                   "exchange_verified": all(v["exchange_verified"] for v in head.values()),
                   "exchange_verified_what": "per collection and rank, untimed: ~16 sampled rows of the rank's block (first, last, seeded "
                                             "places) recomputed by the rank alone -- a dictionary of its own, every pair walked locally -- "
                                             "equal bit for bit to the block after the exchange, all five outputs; conjunction over the ranks "
                                             "(distributed.verify_exchange; per pair the contract is reference src/lib.rs:470-508)",
                   "collection": "50 families of related signatures (SURVEY.md 8d): pairs across families share no hash and are "
                                 "filled without being walked (DESIGN.md 3.4) -- a property of the collection, not of the kernel",
                   "families": head["families"],
                   "one_component": dict(head["one_component"], collection="the same signatures with one hash (a contaminant k-mer) shared by all: one "
                                                                            "connected component; the frequent hash is set aside and pairs that share "
                                                                            "nothing else are decided from per-sketch records (DESIGN.md 3.4)"),
                   "one_family": dict(head["one_family"], collection="ONE family: every pair shares hundreds of hashes, every pair has to be walked -- "
                                                                     "the kernel's own rate"),
                   "roofline": cmp_roof,
                   "note": "'union elements walked' = pairs walked x num: with two full num-sketches the truncated union walk ends "
                           "after exactly num elements (reference src/lib.rs:470-499); per walked pair the effective traffic of "
                           "SURVEY.md 8d is 32 008 B served from LDS/L2, compulsory HBM traffic is N*16 KB in + N^2*8 B out"}
        for n_sig in sizes[1:]:
            compare["block_%d" % n_sig] = dict(per_size[n_sig], workload="BASELINE configs[2]: %d x %d all-vs-all, num=%d" % (n_sig, n_sig, NUM))

    # ---------------------------------------------------------------- host input (PCIe-inclusive, reported beside value)
    host_input = None
    if rank == 0 and world == 1 and args.host_gb > 0:
        nh = min(int(args.host_gb * 1e9), total) // REC_LEN * REC_LEN
        host = seq[:nh].cpu().numpy()                     # pageable host memory, like a caller's buffer
        hoff = np.arange(nh // REC_LEN + 1, dtype=np.uint64) * np.uint64(REC_LEN)
        best = None
        for _ in range(3):
            mhh = pkg.KmerMinHash(0, K, False, 42, MAX_HASH, False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.smh_add_sequences(mhh._p, host.ctypes.data_as(C.c_char_p), hoff.ctypes.data_as(C.POINTER(C.c_uint64)), nh // REC_LEN, True)
            n_ret = len(mhh)
            dth = time.perf_counter() - t0
            assert rc == 0
            best = dth if best is None else min(best, dth)
        ref = pkg.KmerMinHash(0, K, False, 42, MAX_HASH, False)
        ref.add_sequences_dev(seq.data_ptr(), nh, hoff, True, stream)
        assert len(ref) == n_ret and (ref.mins_np() == mhh.mins_np()).all(), "host-input sketch differs from the device-input sketch"
        host_input = {"bytes": nh, "seconds": best, "GB_per_s": nh / best / 1e9,
                      "kmers_per_s": (nh // REC_LEN) * (REC_LEN - K + 1) / best,
                      "what": "the same workload handed over as HOST bytes (pageable memory, what the reference's boundary passes: "
                              "smh_add_sequences = kmerminhash_add_sequence per record): upload in chunks on a second stream while the "
                              "chunks already on the device are hashed.  PCIe-inclusive; never `value`.  Sketch equal to the "
                              "device-input sketch of the same bytes"}
        del host

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        cores = host_cores()
        scale = args.cpu_seconds / 10.0

        def gpu_sketch(first, count):
            g = pkg.KmerMinHash(0, K, False, 42, MAX_HASH, False)
            g.add_sequences_dev(seq.data_ptr() + first * REC_LEN, count * REC_LEN, offsets[: count + 1], True, stream)
            return g.mins_np()

        # sketch, one core
        (r1, p1), = run_workers([["sketch", 0, n_rec, 10.0 * scale]])
        assert (gpu_sketch(0, r1["records"]) == np.load(p1)).all(), "GPU sketch differs from the CPU oracle (1-core sample)"
        one_core = r1["records"] * (REC_LEN - K + 1) / r1["seconds"]
        # sketch, all cores: worker w takes its own stretch of records
        per_w = max(1, min(n_rec // max(1, cores), 200))
        ws = run_workers([["sketch", w * per_w, (w + 1) * per_w, 5.0 * scale] for w in range(cores) if (w + 1) * per_w <= n_rec])
        for w, (r, p) in enumerate(ws):
            assert (gpu_sketch(w * per_w, r["records"]) == np.load(p)).all(), "GPU sketch differs from the CPU oracle (worker %d)" % w
        all_cores = sum(r["records"] for r, _ in ws) * (REC_LEN - K + 1) / max(r["seconds"] for r, _ in ws)
        # BASELINE configs[0] on the CPU, and the same through the GPU path
        (r0, p0), = run_workers([["config0"]])
        g0 = pkg.KmerMinHash(500, K, False, 42, 0, False)
        rec0 = torch.empty(REC_LEN, dtype=torch.uint8, device="cuda")
        assert L.smh_synth_dna_dev(C.c_void_p(rec0.data_ptr()), 0, REC_LEN, 1, 0, C.c_void_p(stream)) == 0
        g0.add_sequences_dev(rec0.data_ptr(), REC_LEN, np.array([0, REC_LEN], dtype=np.uint64), False, stream)
        assert (g0.mins_np() == np.load(p0)).all() and g0.compare(g0) == 1.0 and r0["self_compare"] == 1.0
        cpu = {"value": one_core, "unit": "k-mers/s", "cores": 1, "kind": "port",
               "sample": "first %d of the %d records (1 MB each) of the same workload through the C oracle's add_sequence "
                         "(input generation not timed); sketch checked equal to the GPU's" % (r1["records"], n_rec),
               "all_cores": {"value": all_cores, "cores": len(ws), "host_cores": cores,
                             "sample": "%d independent workers, one stretch of records each (%d records in all), "
                                       "every sketch checked equal to the GPU's" % (len(ws), sum(r["records"] for r, _ in ws))},
               "config0": {"workload": "BASELINE configs[0]: 1 MB synthetic DNA, k=31, num=500, compare to itself",
                           "sketch_kmers_per_s": r0["kmers"] / r0["sketch_seconds"], "compare_pairs_per_s": 1.0 / r0["compare_seconds"],
                           "self_compare": r0["self_compare"], "cores": 1, "gpu_sketch_equal": True}}
        if protein is not None:
            (rp, pp), = run_workers([["protein", 0, len(poffsets) - 1, 5.0 * scale]])
            gp = pkg.KmerMinHash(0, PROT_K, True, 42, MAX_HASH, True)
            gp.add_sequences_dev(pseq.data_ptr(), rp["records"] * REC_LEN, poffsets[: rp["records"] + 1], True, stream)
            ref = np.load(pp)
            assert (gp.mins_np() == ref[0]).all() and (gp.abunds_np() == ref[1]).all(), "GPU protein sketch differs from the CPU oracle"
            ppw = max(1, min((len(poffsets) - 1) // max(1, cores), 400))
            pw = run_workers([["protein", w * ppw, (w + 1) * ppw, 4.0 * scale] for w in range(cores) if (w + 1) * ppw <= len(poffsets) - 1])
            for w, (r, p_) in enumerate(pw):
                gw = pkg.KmerMinHash(0, PROT_K, True, 42, MAX_HASH, True)
                gw.add_sequences_dev(pseq.data_ptr() + w * ppw * REC_LEN, r["records"] * REC_LEN, poffsets[: r["records"] + 1], True, stream)
                refw = np.load(p_)
                assert (gw.mins_np() == refw[0]).all() and (gw.abunds_np() == refw[1]).all(), "GPU protein sketch differs from the CPU oracle (worker %d)" % w
            protein["cpu_baseline"] = {
                "value": rp["records"] * PROT_WINDOWS_PER_REC / rp["seconds"], "unit": "windows/s", "cores": 1, "kind": "port",
                "sample": "first %d record(s) (1 MB each) of the same workload through the C oracle's protein arm (to_aa on six frames + "
                          "add_word per window, reference src/lib.rs:275-302); hashes and abundances equal to the GPU's" % rp["records"],
                "all_cores": {"value": sum(r["records"] for r, _ in pw) * PROT_WINDOWS_PER_REC / max(r["seconds"] for r, _ in pw),
                              "cores": len(pw), "host_cores": cores,
                              "sample": "%d workers, one stretch of records each (%d records in all); every sketch equal to the GPU's"
                                        % (len(pw), sum(r["records"] for r, _ in pw))}}
        if compare is not None and cmp_outs:
            tmpd = tempfile.mkdtemp(prefix="smh_sigs_")
            TMP_DIRS.append(tmpd)
            for n_sig in sorted(cmp_outs):
                # the CPU sample: every row of the 1 000 x 1 000 block (bounded by the budget); of the 10 000 x 10 000 matrix its
                # first 1 000-row block -- all of it on all cores for the family collection, a few seconds'
                # worth for the other two
                block = min(n_sig, 1000)
                for key, kind_id in (("families", 0), ("one_component", 1), ("one_family", 2)):
                    gj = cmp_outs[n_sig][key]
                    path = os.path.join(tmpd, "sigs%d_%d.npy" % (n_sig, kind_id))
                    np.save(path, collection(kind_id, 0, n_sig))
                    (rc1, pc1), = run_workers([["compare", n_sig, 0, block, 2.0 * scale, kind_id, path]])
                    j1 = np.load(pc1)
                    assert (gj[: j1.shape[0]] == j1).all(), "GPU matrix differs from the CPU oracle (%s, 1-core rows)" % key
                    per_w = max(1, block // cores)
                    budget = (24.0 if (n_sig > 1000 and kind_id == 0) else 2.0) * scale
                    cw = run_workers([["compare", n_sig, w * per_w, (w + 1) * per_w, budget, kind_id, path] for w in range(cores)
                                      if (w + 1) * per_w <= block])
                    for w, (r, p) in enumerate(cw):
                        jw = np.load(p)
                        assert (gj[w * per_w: w * per_w + jw.shape[0]] == jw).all(), "GPU matrix differs from the CPU oracle (%s, worker %d)" % (key, w)
                    target = compare[key] if n_sig == sizes[0] else compare["block_%d" % n_sig][key]
                    target["cpu_baseline"] = {
                        "value": rc1["rows"] * n_sig / rc1["seconds"], "unit": "pairs/s", "cores": 1, "kind": "port", "n_signatures": n_sig,
                        "sample": "rows 0..%d x all %d columns through the C oracle's compare (two merges + two intersections per "
                                  "pair, reference src/lib.rs:470-508); equal to the GPU's rows" % (rc1["rows"] - 1, n_sig),
                        "all_cores": {"value": sum(r["rows"] for r, _ in cw) * n_sig / max(r["seconds"] for r, _ in cw),
                                      "cores": len(cw), "host_cores": cores,
                                      "sample": "%d workers, one stretch of rows of the first %d-row block each (%d rows x %d columns in all), "
                                                "equal to the GPU's rows" % (len(cw), block, sum(r["rows"] for r, _ in cw), n_sig)}}
                    os.remove(path)

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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_dna_rolling<31,...>", "kernel_ms_avg": kern_ms, "launches_per_step": launches_per_step,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu_bound": valu_bound,
                         "note": "1 B per k-mer: the path is integer-VALU bound, not HBM bound (SURVEY.md 8d; DESIGN.md 'Roofline'); "
                                 "valu_bound prices the kernel against the guide's VALU issue peak"},
            "cpu_baseline": cpu,
            "host_input": host_input,
            "union_across_ranks": union,
            "protein": protein,
            "compare": compare,
        }
        print(json.dumps(line))
        sys.stdout.flush()
    import shutil
    for d in TMP_DIRS:
        shutil.rmtree(d, ignore_errors=True)
    if world > 1:
        dist.destroy_process_group()
    # a matrix that fails its own check is not a result: the line above says which check, the exit code says so too
    if compare is not None and rank == 0 and not (compare["exchange_verified"] and compare["self_jaccard_is_1"]):
        sys.exit(3)
    if union is not None and rank == 0 and not union["verified"]:
        sys.exit(3)
    if protein is not None and protein["union_across_ranks"] is not None and rank == 0 and not protein["union_across_ranks"]["verified"]:
        sys.exit(3)


if __name__ == "__main__":
    main()
