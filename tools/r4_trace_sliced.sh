cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r4_trace_sliced
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o t -- python3 $R/tools/project_sharded.py 10000 ${1:-one_family} 8 > $R/gpurun_out/r4_trace_sliced.log 2>&1
tail -1 $R/gpurun_out/r4_trace_sliced.log | cut -c1-200
db=$(find $out -name "*.db" | head -1)
python3 $R/tools/trace_db.py $db k_slice_parts > $R/gpurun_out/r4_launches_sliced_slice.txt 2>&1
python3 $R/tools/trace_db.py $db k_reassemble > $R/gpurun_out/r4_launches_sliced_compare.txt 2>&1
tail -1 $R/gpurun_out/r4_launches_sliced_slice.txt; tail -1 $R/gpurun_out/r4_launches_sliced_compare.txt
