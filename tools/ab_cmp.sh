# A/B of compare-kernel variants: dense and sparse matrices per library under sourmash-rust_amd/lib*/, then the parity tests
for d in sourmash-rust_amd/lib sourmash-rust_amd/lib_v*; do
  [ -f $d/libsourmash_amd.so ] || continue
  export SOURMASH_AMD_LIB=$PWD/$d/libsourmash_amd.so
  echo "== $d"
  for m in "1000 families" "10000 families" "10000 all_tiles" "2500 all_tiles"; do timeout -k 10 120 python tools/prof_compare_1000.py $m 2>&1 | tail -1 | cut -c1-90; done
  timeout -k 10 500 python -m pytest tests/test_gpu_compare.py -x -q 2>&1 | tail -1
done
