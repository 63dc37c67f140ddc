# small blocks, tiled route forced: which tile height / kernel (see sweep_pf.sh)
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
export PROF_FORCE_TILED=1
for n in 100 200 300 450 600; do
  for cfg in "1 0" "1 1" "2 0" "2 1" "4 1"; do
    set -- $cfg
    echo "N=$n rows=$(($1*4)) pf=$2 :: $(SOURMASH_AMD_CMP_GEO=$1,4,8 SOURMASH_AMD_CMP_PF=$2 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 8 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-30)"
  done
done
