# packed (2 bits per base) against byte tile of k_dna_rolling, and the packed kernel's variants: experiments build, sketch step only
export SOURMASH_AMD_LIB=$PWD/sourmash-rust_amd/lib_vexp/libsourmash_amd.so
run() { python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-compare --host-gb 0 --protein-gb 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f ms/step, kernel %.2f ms, %.1f G k-mers/s' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['value']/1e9))"; }
echo "byte tile (4 waves/SIMD): $(SOURMASH_AMD_DNA_PK=0 run)"
echo "packed, default (6,1)   : $(run)"
for v in 8,2 8,1 7,1 6,1 5,1 4,1 6,2; do echo "packed, minw,hb = $v    : $(SOURMASH_AMD_DNA_PKV=$v run)"; done
echo "byte tile (4 waves/SIMD): $(SOURMASH_AMD_DNA_PK=0 run)"
