# Pipelined (k_compare_tiled_pf) against plain tiled kernel per tile height and block size, ONE family (every pair walked).
# Needs the experiments build: make -C sourmash-rust_amd/csrc EXTRA=-DSMH_EXPERIMENTS OUT=../lib_vexp OBJ=../build_vexp
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for n in ${SWEEP_N:-500 700 1000 1500 2500 4000 10000}; do
  for rpw in 4 2 1; do
    for pf in 0 1; do
      if [ $n -ge 10000 ] && [ $rpw -eq 1 ]; then continue; fi
      echo "N=$n rows=$((rpw*4)) pf=$pf :: $(SOURMASH_AMD_CMP_GEO=$rpw,4,8 SOURMASH_AMD_CMP_PF=$pf timeout -k 10 120 python tools/prof_compare_1000.py $n one_family 6 2>/dev/null | tail -1 | sed -e 's/.*kernels ms//' | cut -c1-60)"
    done
  done
done
