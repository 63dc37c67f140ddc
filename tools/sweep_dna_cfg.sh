# geometry sweep of the rolling kernel (needs lib_vx built with -DSMH_EXPERIMENTS)
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vx/libsourmash_amd.so
for cfg in 512,7,2 512,7,4 512,7,1 512,6,2 512,6,4 256,7,2 256,7,4 256,6,2 256,6,4 512,5,2 256,5,4; do
  r=$(SOURMASH_AMD_DNA_CFG=$cfg timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-compare --cpu-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.1f G/s  step %.2f ms  kernel %.2f ms' % (d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms_avg']))")
  echo "cfg $cfg: $r"
done
