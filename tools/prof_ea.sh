# Memory-side (fabric) read requests of the L2 by size -- TCC_EA0_RDREQ with its 32 / 64 / 128-byte classes -- for the
# two sketch kernels: exact bytes, where FETCH_SIZE (requests x 64 B) needs the guide's x2 for 128-byte requests and
# over-counts anything smaller.  Run from the repo root.
out=$PWD/gpurun_out/${1:-ea}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
ONE="python3 $PWD/bench.py --gpus 1 --steps 1 --warmup 0 --cpu-seconds 0 --no-compare"
C5="python3 $PWD/tools/bench_c5.py 12500"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $out/dna_rd -- $ONE > $out/dna_rd.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $out/c5_rd -- $C5 > $out/c5_rd.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for tag, kern, inp in (("dna", "k_dna_rolling", 10.0e9), ("c5", "k_protein_fused", 37.6e9)):
    acc = collections.defaultdict(float)
    for f in glob.glob("%s/%s_rd/**/*counter_collection.csv" % (out, tag), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    print(tag, dict(acc))
    n, n32, n64, n128 = (acc.get("TCC_EA0_RDREQ_sum", 0), acc.get("TCC_EA0_RDREQ_32B_sum", 0), acc.get("TCC_EA0_RDREQ_64B_sum", 0), acc.get("TCC_EA0_RDREQ_128B_sum", 0))
    b = 32 * n32 + 64 * n64 + 128 * n128
    print("%s: %.0f read requests = %.0f x 32 B + %.0f x 64 B + %.0f x 128 B = %.3f GB = %.4f x the %.1f GB of input" % (tag, n, n32, n64, n128, b / 1e9, b / inp, inp / 1e9))
PY
tail -3 $out/c5_rd.log
