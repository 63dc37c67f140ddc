# A/B of protein-kernel variants: C5 share timing + parity tests per library under sourmash-rust_amd/lib*/
for d in sourmash-rust_amd/lib sourmash-rust_amd/lib_v*; do
  [ -f $d/libsourmash_amd.so ] || continue
  export SOURMASH_AMD_LIB=$PWD/$d/libsourmash_amd.so
  r=$(timeout -k 10 300 python tools/bench_c5.py 12500 2>&1 | grep -E "C5 share|kernel protein_fused" | tr '\n' ' ')
  t=$(timeout -k 10 400 python -m pytest tests/test_gpu_sketch.py -x -q -k "protein or grouped or random or nul" 2>&1 | tail -1)
  echo "$d: $r | $t"
done
