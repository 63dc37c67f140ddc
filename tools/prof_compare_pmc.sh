# Round-3 counters of the compare kernels (GPU box, from the repo root): PMC passes in runs of their own (no trace
# domains beside --kernel-trace), one workload per directory under gpurun_out/$1/.
out=$PWD/gpurun_out/${1:-cmp_pmc}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
cd /tmp
run() {   # name, command...
  name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/$name/sq -- "$@" > $out/$name.sq.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/$name/sq2 -- "$@" > $out/$name.sq2.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$name/fetch -- "$@" > $out/$name.fetch.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/$name/write -- "$@" > $out/$name.write.log 2>&1
  ( echo "== $name: $@"; tail -1 $out/$name.sq.log; for p in sq sq2 fetch write; do python3 $R/tools/pmc_summary.py $out/$name/$p k_compare_ k_fill_disjoint; done ) > $out/$name.summary.txt 2>&1
}
run tiled_10000_dense python3 $R/tools/prof_compare_1000.py 10000 one_family 3
run tiled_10000_families python3 $R/tools/prof_compare_1000.py 10000 families 3
run tiled_1000_dense python3 $R/tools/prof_compare_1000.py 1000 one_family 4
run comp_1000_families python3 $R/tools/prof_compare_1000.py 1000 families 4
run comp_1000_dense python3 $R/tools/prof_compare_1000.py 1000 components 4
run few_index python3 $R/tools/bench_index.py
cd $R
cat $out/*.summary.txt
