# Issue-side PMC passes on the 10 GB DNA step (k_dna_rolling): where the wave cycles go.  Run from the repo root.
out=$PWD/gpurun_out/${1:-dnaissue}; mkdir -p $out
export TMPDIR=/tmp PYTHONPATH=$PWD
R=$PWD
cd /tmp
CMD="python3 $R/bench.py --gpus 1 --steps 1 --warmup 0 --cpu-seconds 0 --no-compare --host-gb 0 --protein-gb 0"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $out/p1 -- $CMD > $out/p1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $out/p2 -- $CMD > $out/p2.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM --kernel-trace --output-format csv -d $out/p3 -- $CMD > $out/p3.log 2>&1
cd $R
for p in p1 p2 p3; do python3 tools/pmc_summary.py $out/$p k_dna_rolling; done
