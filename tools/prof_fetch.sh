# FETCH_SIZE / WRITE_SIZE passes only (DNA bench step and the C5 share).  Run from the repo root.
out=$PWD/gpurun_out/${1:-fetch}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
ONE="python3 $PWD/bench.py --gpus 1 --steps 1 --warmup 0 --cpu-seconds 0 --no-compare"
C5="python3 $PWD/tools/bench_c5.py 12500"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/dna_fetch -- $ONE > $out/dna_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/dna_write -- $ONE > $out/dna_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/c5_fetch -- $C5 > $out/c5_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/c5_write -- $C5 > $out/c5_write.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for tag, kern, inp in (("dna", "k_dna_rolling", 10.0e9), ("c5", "k_protein_fused", 37.6e9)):
    tot = {}
    for c in ("fetch", "write"):
        v = 0.0
        for f in glob.glob("%s/%s_%s/**/*counter_collection.csv" % (out, tag, c), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    v += float(r["Counter_Value"])
        tot[c] = v
    b = tot["fetch"] * 2 * 1024 + tot["write"] * 1024
    print("%s: FETCH_SIZE %.0f KB x2 + WRITE_SIZE %.0f KB = %.3f GB = %.4f x the %.1f GB of input" % (tag, tot["fetch"], tot["write"], b / 1e9, b / inp, inp / 1e9))
PY
