# Round-3 re-sweep of the tiled kernel's staging geometry after the arithmetic merge walk (experiments build in lib_vexp):
# pooled elements per sketch per range x LDS caps (row pool dwords, column elements per range).
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for cfg in "24 1024,48" "32 1280,60" "40 1536,72" "48 1792,88" "64 2304,112" "16 768,36" "32 1024,48" "24 1280,60"; do
  set -- $cfg
  for m in "10000 one_family 5" "1000 one_family 8" "10000 families 5"; do
    echo "per_range=$1 lds=$2 :: $(SOURMASH_AMD_CMP_PER_RANGE=$1 SOURMASH_AMD_CMP_LDS=$2 timeout -k 10 120 python tools/prof_compare_1000.py $m 2>/dev/null | tail -1 | cut -c1-140)"
  done
done
