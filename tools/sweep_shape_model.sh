# chain time of one tile per height, and what partial rounds cost: forced 8 / 16 / 32-row pipelined tiles, ONE family, tiled route forced
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
export PROF_FORCE_TILED=1
for n in 300 600 900 1000 1200 1400 1600 1800 2000 2300; do
  for rpw in 2 4 8; do
    echo "N=$n rows=$((rpw*4)) :: $(SOURMASH_AMD_CMP_GEO=$rpw,4,8 SOURMASH_AMD_CMP_PF=1 timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 10 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' -e "s/'compare_comp.*'tiles_visited'/ 'tiles'/" -e "s/, 'tiles_total.*//" | cut -c1-70)"
  done
done
