# A/B of variant libraries on the one-genome-per-call latency at three genome sizes
for d in sourmash-rust_amd/lib_v*; do
  [ -f $d/libsourmash_amd.so ] || continue
  export SOURMASH_AMD_LIB=$PWD/$d/libsourmash_amd.so
  for n in 1000000 5000000 20000000; do
    timeout -k 10 200 python tools/bench_one_genome.py $n 2>&1 | grep -v amdgpu | sed "s|^|$d: |"
  done
done
