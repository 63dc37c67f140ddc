export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for n in 4000 10000; do
  echo "N=$n plain16 :: $(SOURMASH_AMD_CMP_GEO=4,4,8 SOURMASH_AMD_CMP_PF=0 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 6 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-40)"
  echo "N=$n pf32 capA x2 :: $(SOURMASH_AMD_CMP_PF32_BIG=1 SOURMASH_AMD_CMP_GEO=8,4,8 SOURMASH_AMD_CMP_PF=1 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 6 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-40)"
  echo "N=$n pf32 capA x1 :: $(SOURMASH_AMD_CMP_GEO=8,4,8 SOURMASH_AMD_CMP_PF=1 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 6 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' -e 's/compare_comp.*lds_overflow/ lds_overflow/' | cut -c1-120)"
done
