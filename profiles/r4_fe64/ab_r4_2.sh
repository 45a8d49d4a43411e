# round 4, second GPU call: the shortened tail (am_k_gather_wg, am_k_refine_late, ticket inside the extraction kernel) against
# round 3's library, the ring-capacity variant of am_k_fe3, kernel stats at both densities and at 20 / 2 Msps
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" "FE=3 LIB=build/var/lib_cr6.so" 2>&1 | tee gpurun_out/ab_r4_2.txt
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" 2>&1 | tee -a gpurun_out/ab_r4_2.txt
BENCH_ARGS="--workload 20msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" 2>&1 | tee -a gpurun_out/ab_r4_2.txt
BENCH_ARGS="--workload 2msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" 2>&1 | tee -a gpurun_out/ab_r4_2.txt
STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | tee gpurun_out/kstats_r4_2.txt
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | tee gpurun_out/kstats_r4_2_l2000.txt
timeout 300 python bench.py --steps 12 --warmup 3 > gpurun_out/bench_r4_2.json 2> gpurun_out/bench_r4_2.err; tail -c 1500 gpurun_out/bench_r4_2.json
