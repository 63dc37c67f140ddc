# four-pass + tie fix (default) against the full eight-pass sort of the pooled hashes, whole call, no profiler
for cfg in "1000 one_family" "1000 families" "1000 one_component" "3000 families" "10000 families" "10000 one_family"; do
  set -- $cfg
  echo "4-pass : $(python tools/prof_compare_1000.py $1 $2 14 2>/dev/null | tail -1 | cut -c1-150)"
  echo "8-pass : $(PROF_DICT=full python tools/prof_compare_1000.py $1 $2 14 2>/dev/null | tail -1 | cut -c1-150)"
done
