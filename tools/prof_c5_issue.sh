# Issue-side PMC passes on the C5 share (protein arm): where the wave cycles go.  Run from the repo root.
out=$PWD/gpurun_out/${1:-c5issue}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
CMD="python3 $PWD/tools/bench_c5.py 12500"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/p1 -- $CMD > $out/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/p2 -- $CMD > $out/p2.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for p in ("p1", "p2"):
    acc = collections.defaultdict(float)
    for f in glob.glob(out + "/" + p + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_protein_fused" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
    for k in sorted(acc):
        print("%-28s %16.0f   per 64 positions %10.2f" % (k, acc[k], acc[k] / (37.6e9 / 64)))
PY
