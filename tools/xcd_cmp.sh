for mode in "" "SOURMASH_AMD_CMP_NO_XCD=1"; do
  for all in "" "SOURMASH_AMD_CMP_ALL_TILES=1"; do
    echo "[$mode $all] $(env $mode $all timeout -k 10 200 python tools/prof_compare_1000.py 10000 2>/dev/null | tail -1)"
  done
done
