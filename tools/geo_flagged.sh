for g in "" 4,4,8 2,4,8 1,4,8 1,8,8 2,8,8; do
  e=""; [ -n "$g" ] && e="SOURMASH_AMD_CMP_GEO=$g"
  for n in 2500 10000; do
    echo "GEO [$g] $(env $e timeout -k 10 120 python tools/prof_compare_1000.py $n 2>/dev/null | tail -1)"
  done
done
