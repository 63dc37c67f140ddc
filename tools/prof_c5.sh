# PMC passes on the C5 share (protein arm, 12.5 GB): SQ counters, FETCH_SIZE, WRITE_SIZE, kernel stats.  Run from the repo root.
out=$PWD/gpurun_out/${1:-c5prof}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
CMD="python3 $PWD/tools/bench_c5.py 12500"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $CMD > $out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out/sq -- $CMD > $out/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -- $CMD > $out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -- $CMD > $out/write.log 2>&1
grep -E "C5 share|kernel protein" $out/stats.log
