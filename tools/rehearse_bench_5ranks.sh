BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo timeout -k 10 700 python bench.py --gpus 5 --gb 0.2 --steps 1 --warmup 1 --cpu-seconds 0 > gpurun_out/r03_bench_5ranks.json 2> gpurun_out/r03_bench_5ranks.err
tail -c 300 gpurun_out/r03_bench_5ranks.err
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_5ranks.json") if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["union_across_ranks"]["hashes"], d["union_across_ranks"]["union_ms"])
c=d["compare"]
for k in ("families","one_component","one_family"):
    v=c[k]; print(k, round(v["seconds"]*1e3,2), v["self_jaccard_is_1"], v["route"], {a:round(b,2) for a,b in v["rank0_phase_ms"].items()})
PY
