# Counters of the pipelined tiled kernel (k_compare_tiled_pf) on the blocks the plan gives it: dense 1 000 x 1 000 (8-row tiles),
# dense 2 500 x 2 500 (16-row tiles), the family collection at 10 000 x 10 000 (16-row tiles).  Same passes as prof_compare_pmc.sh.
out=$PWD/gpurun_out/${1:-cmp_pmc_pf}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
cd /tmp
run() {   # name, command...
  name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/$name/sq -- "$@" > $out/$name.sq.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/$name/sq2 -- "$@" > $out/$name.sq2.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$name/fetch -- "$@" > $out/$name.fetch.log 2>&1
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/$name/write -- "$@" > $out/$name.write.log 2>&1
  ( echo "== $name: $@"; tail -1 $out/$name.sq.log; for p in sq sq2 fetch write; do python3 $R/tools/pmc_summary.py $out/$name/$p k_compare_tiled; done ) > $out/$name.summary.txt 2>&1
}
run pf_1000_dense python3 $R/tools/prof_compare_1000.py 1000 one_family 4
run pf_2500_dense python3 $R/tools/prof_compare_1000.py 2500 one_family 4
run pf_10000_families python3 $R/tools/prof_compare_1000.py 10000 families 3
cd $R
cat $out/*.summary.txt
