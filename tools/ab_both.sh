# A/B of sketch-kernel variants (libraries under sourmash-rust_amd/lib_v*/): DNA bench + C5 share + sketch parity tests.
for d in sourmash-rust_amd/lib_v*; do
  [ -f $d/libsourmash_amd.so ] || continue
  export SOURMASH_AMD_LIB=$PWD/$d/libsourmash_amd.so
  r=$(timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-compare --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DNA step %.2f ms kernel %.2f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_avg']))")
  c=$(timeout -k 10 300 python tools/bench_c5.py 12500 2>&1 | grep -E "kernel protein_fused" | tr '\n' ' ')
  t=$(timeout -k 10 400 python -m pytest tests/test_gpu_sketch.py -x -q 2>&1 | tail -1)
  echo "$d: $r | $c | $t"
done
