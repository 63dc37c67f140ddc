# A/B of rolling-kernel variants: each library under sourmash-rust_amd/lib*/ sketches the benchmark's 10 GB and must
# pass the sketch parity tests.  Usage (GPU box): bash tools/ab_dna.sh
for d in sourmash-rust_amd/lib sourmash-rust_amd/lib_v*; do
  [ -f $d/libsourmash_amd.so ] || continue
  export SOURMASH_AMD_LIB=$PWD/$d/libsourmash_amd.so
  r=$(timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-compare --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f G/s  step %.2f ms  kernel %.2f ms' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms_avg']))")
  t=$(timeout -k 10 400 python -m pytest tests/test_gpu_sketch.py -x -q 2>&1 | tail -1)
  echo "$d: $r | $t"
done
