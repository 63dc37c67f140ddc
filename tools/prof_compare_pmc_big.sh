# Counters of the tiled kernel on the dense 10 000 x 10 000 block (32-row pipelined tiles): the figures behind profiles/r03_pmc_compare_tiled.json
# and bench.py's compare.roofline.  PMC passes in runs of their own (see prof_compare_pmc.sh).
out=$PWD/gpurun_out/${1:-cmp_pmc_big}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
cd /tmp
name=tiled_10000_dense
cmd="python3 $R/tools/prof_compare_1000.py 10000 one_family 3"
timeout -k 10 280 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/$name/sq -- $cmd > $out/$name.sq.log 2>&1
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $out/$name/sq2 -- $cmd > $out/$name.sq2.log 2>&1
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/$name/fetch -- $cmd > $out/$name.fetch.log 2>&1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/$name/write -- $cmd > $out/$name.write.log 2>&1
( echo "== $name: $cmd"; tail -1 $out/$name.sq.log; for p in sq sq2 fetch write; do python3 $R/tools/pmc_summary.py $out/$name/$p "k_compare_tiled_pf<false, 4"; done ) > $out/$name.summary.txt 2>&1
cd $R
cat $out/$name.summary.txt
