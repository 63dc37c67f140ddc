# End-of-round evidence (GPU box, from the repo root): the default bench line, the kernel-trace summary of the same build and
# workload, the launch sequences of one compare call (world 1 and one rank of eight), the compare calls and the sharded phases.
R=$PWD
mkdir -p gpurun_out/r4_final
python bench.py > gpurun_out/r4_final/bench_line.json 2> gpurun_out/r4_final/bench.err
tail -c 600 gpurun_out/r4_final/bench_line.json; echo
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_final/stats -- python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --cpu-seconds 0 --host-gb 0 > $R/gpurun_out/r4_final/stats.log 2>&1)
find gpurun_out/r4_final/stats -name "*kernel_stats.csv" | head -2
bash tools/r4_trace.sh > gpurun_out/r4_final/trace.log 2>&1
bash tools/r4_trace_sliced.sh one_family > gpurun_out/r4_final/trace_sliced.log 2>&1
(for n in 1000 10000; do for m in families one_component one_family; do python tools/prof_compare_1000.py $n $m 12 2>&1 | grep "^n="; done; done) > gpurun_out/r04_compare_calls.txt
(echo "# round 4, final build of the round: python tools/project_sharded.py 10000 <collection> 1 2 4 8"; python tools/project_sharded.py 10000 one_family 1 2 4 8 2>&1 | grep "^N="; python tools/project_sharded.py 10000 families 1 2 4 8 2>&1 | grep "^N=") > gpurun_out/r04_sharded_compute_phases.txt
echo final done
