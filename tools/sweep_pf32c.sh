export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for n in 1500 2000 2500 3200; do
  echo "N=$n pf16 :: $(SOURMASH_AMD_CMP_GEO=4,4,8 SOURMASH_AMD_CMP_PF=1 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 8 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-40)"
  echo "N=$n pf32 :: $(SOURMASH_AMD_CMP_GEO=8,4,8 SOURMASH_AMD_CMP_PF=1 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 8 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-40)"
done
for cfg in "4,4,8 0" "8,4,8 1"; do
  set -- $cfg
  echo "ranks geo=$1 pf=$2 :: $(SOURMASH_AMD_CMP_GEO=$1 SOURMASH_AMD_CMP_PF=$2 timeout -k 10 200 python tools/project_sharded.py 10000 one_family 2 4 8 2>/dev/null | grep '^N=' | sed -e 's/compute critical path//' -e 's/slice.*compare/compare/' -e 's/exchange.*//' | tr '\n' ';')"
done
