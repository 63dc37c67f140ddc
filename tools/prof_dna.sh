# Evidence for the hot kernel (GPU box, from the repo root): kernel-trace stats, then PMC passes in
# runs of their own (SQ counters; FETCH_SIZE; WRITE_SIZE).  Output under gpurun_out/$1/.
out=gpurun_out/${1:-prof}; mkdir -p $out
R=$PWD
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --cpu-seconds 0 --no-compare --host-gb 0 --protein-gb 0"
ONE="python3 $R/bench.py --gpus 1 --steps 1 --warmup 0 --cpu-seconds 0 --no-compare --host-gb 0 --protein-gb 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- $BENCH > $R/$out/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $R/$out/sq -- $ONE > $R/$out/sq.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$out/fetch -- $ONE > $R/$out/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$out/write -- $ONE > $R/$out/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$out/grbm -- $ONE > $R/$out/grbm.log 2>&1
cd $R
find $out -name "*.csv" | head -40
