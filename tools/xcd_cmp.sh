# XCD-aware tile order on/off (needs a library built with -DSMH_EXPERIMENTS: make -C sourmash-rust_amd/csrc EXTRA=-DSMH_EXPERIMENTS)
for mode in "" "SOURMASH_AMD_CMP_NO_XCD=1"; do
  for all in families all_tiles; do
    echo "[$mode $all] $(env $mode timeout -k 10 200 python tools/prof_compare_1000.py 10000 $all 2>/dev/null | tail -1)"
  done
done
