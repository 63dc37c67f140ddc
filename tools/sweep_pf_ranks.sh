export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
for pf in 0 1 0 1; do
  echo "pf=$pf :: $(SOURMASH_AMD_CMP_GEO=4,4,8 SOURMASH_AMD_CMP_PF=$pf timeout -k 10 200 python tools/project_sharded.py 10000 one_family 4 8 2>/dev/null | cut -c1-190)"
done
