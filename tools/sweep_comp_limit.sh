for lim in 262144 131072 65536 32768; do
  echo "limit=$lim :: w8 fam: $(PROF_COMP_LIMIT=$lim python tools/project_sharded.py 10000 families 8 2>/dev/null | grep '^N=' | sed -e 's/.*compare max/compare max/' -e 's/exchange.*rank 0:/| rank 0:/' | cut -c1-120)"
  for cfg in "1000 families" "2000 families" "3000 families" "1000 one_component"; do
    echo "   limit=$lim $cfg: $(PROF_COMP_LIMIT=$lim python tools/prof_compare_1000.py $cfg 12 2>/dev/null | tail -1 | sed -e 's/ per matrix.*kernels ms/ kernels/' -e "s/'rows_per_tile.*//" | cut -c1-170)"
  done
done
