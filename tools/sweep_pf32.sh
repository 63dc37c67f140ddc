export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for n in 2500 4000 10000; do
  for cfg in "4 0" "4 1" "8 1"; do
    set -- $cfg
    echo "N=$n rows=$(($1*4)) pf=$2 :: $(SOURMASH_AMD_CMP_GEO=$1,4,8 SOURMASH_AMD_CMP_PF=$2 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 6 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-40)"
  done
done
