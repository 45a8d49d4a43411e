# One round's evidence, on the GPU box:   gpurun -- 'bash tools/gpu_profile_round.sh'
#   plain bench (the numbers), three rocprofv3 passes of the same command (stats, FETCH_SIZE, WRITE_SIZE),
#   the phase-clock stamps of the dominant kernel.  Everything lands in gpurun_out/round/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
STEPS=10 bash tools/gpu_prof.sh > $OUT/summary.txt 2>&1
cp gpurun_out/prof_stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
cp gpurun_out/bench_prof.json $OUT/bench_under_rocprof.json
AIRMODES_FE2_CLOCK=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined 2>&1 | grep "fe2 clock" | tail -2 > $OUT/fe2_phase_clocks.txt
python bench.py --workload 2msps --no-cpu-baseline > $OUT/bench_2msps.json 2>/dev/null
python bench.py --workload 20msps --no-cpu-baseline > $OUT/bench_20msps.json 2>/dev/null
tail -c 600 $OUT/bench.json; echo; cat $OUT/fe2_phase_clocks.txt; grep -E "fe2|energy|cand" $OUT/summary.txt | head -8
