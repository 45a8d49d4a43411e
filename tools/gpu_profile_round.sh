# One round's evidence, on the GPU box:   gpurun -- 'bash tools/gpu_profile_round.sh'
#   plain bench (the numbers), three rocprofv3 passes of the same command (stats, FETCH_SIZE, WRITE_SIZE) at the
#   default (stress) density and the PMC passes again at the realistic one, the phase clocks of the dominant kernel
#   (profiling build build/var/lib_fe3prof.so, made by tools/build_variants.sh).  Everything lands in gpurun_out/round/.
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
STEPS=10 timeout 300 bash tools/gpu_prof.sh > $OUT/summary.txt 2>&1
cp gpurun_out/prof_stats/bench_kernel_stats.csv $OUT/kernel_stats.csv
cp gpurun_out/bench_prof.json $OUT/bench_under_rocprof.json
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_prof.sh > $OUT/summary_lambda2000.txt 2>&1
cp gpurun_out/prof_stats/bench_kernel_stats.csv $OUT/kernel_stats_lambda2000.csv
if [ -f build/var/lib_fe3prof.so ]; then
  AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "^fe3" | tail -30 > $OUT/fe3_phase_clocks.txt
fi
if [ -f build/var/lib_rsprof.so ]; then
  # am_k_refine_seg: us per phase and workgroup + the launch's timeline (profiling build), both densities
  AIRMODES_HIP_LIB=$PWD/build/var/lib_rsprof.so timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra 2>&1 >/dev/null | grep "^rseg" | tail -4 > $OUT/refine_seg_phase_clocks.txt
  AIRMODES_HIP_LIB=$PWD/build/var/lib_rsprof.so timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra --lambda 2000 2>&1 >/dev/null | grep "^rseg" | tail -4 >> $OUT/refine_seg_phase_clocks.txt
fi
AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so AIRMODES_FE=2 timeout 120 python bench.py --no-cpu-baseline --no-extra > $OUT/bench_tile_kernel.json 2>/dev/null
timeout 120 python bench.py --force-sharded --no-cpu-baseline --no-extra > $OUT/bench_force_sharded.json 2>/dev/null
timeout 200 python bench.py --force-sharded --backend nccl --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > $OUT/bench_force_sharded_nccl_world1.json
timeout 200 python bench.py --force-sharded --backend nccl --steps 200 --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > $OUT/bench_force_sharded_nccl_world1_200_steps.json
timeout 200 python bench.py --force-sharded --backend nccl --no-steps-in-flight --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > $OUT/bench_force_sharded_nccl_world1_one_step_at_a_time.json
timeout 200 python bench.py --force-sharded --backend nccl --no-steps-in-flight --no-lookahead --no-cpu-baseline --no-extra 2>/dev/null | grep "^{" > $OUT/bench_force_sharded_nccl_world1_two_collectives.json
timeout 400 python bench.py --workload 2msps --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_2msps.json
timeout 400 python bench.py --workload 20msps --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_20msps.json
timeout 200 bash tools/gpu_pmc.sh > $OUT/sq_counters.txt 2>&1
KFILTER=refine_seg timeout 200 bash tools/gpu_pmc.sh > $OUT/sq_counters_refine_seg.txt 2>&1
for w in 20msps 2msps; do BENCH_ARGS="--workload $w" STEPS=10 timeout 200 bash tools/gpu_kstats.sh > $OUT/kernel_stats_$w.txt 2>&1; done
tail -c 900 $OUT/bench.json; echo; cat $OUT/fe3_phase_clocks.txt; grep -E "fe3|fe2|energy|cand|extract" $OUT/summary.txt | head -12
