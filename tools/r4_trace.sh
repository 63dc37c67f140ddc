cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1000 one_family" "1000 families" "10000 families"; do
  set -- $cfg
  out=$R/gpurun_out/r4_trace_$1_$2
  rm -rf $out
  rocprofv3 --kernel-trace --stats -d $out -o t -- python3 $R/tools/prof_compare_1000.py $1 $2 8 > $R/gpurun_out/r4_trace_$1_$2.log 2>&1
  grep "^n=" $R/gpurun_out/r4_trace_$1_$2.log
  db=$(find $out -name "*.db" | head -1)
  python3 $R/tools/trace_db.py $db k_key_span > $R/gpurun_out/r4_launches_$1_$2.txt 2>&1
  tail -1 $R/gpurun_out/r4_launches_$1_$2.txt
done
