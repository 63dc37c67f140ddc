# range masks (default) against the walk from the first range on, whole call, no profiler
for cfg in "1000 one_family" "1000 families" "3000 families" "10000 families" "10000 one_component" "10000 one_family"; do
  set -- $cfg
  echo "masks   : $(python tools/prof_compare_1000.py $1 $2 10 2>/dev/null | tail -1 | cut -c1-150)"
  echo "no masks: $(PROF_NO_MASKS=1 python tools/prof_compare_1000.py $1 $2 10 2>/dev/null | tail -1 | cut -c1-150)"
done
