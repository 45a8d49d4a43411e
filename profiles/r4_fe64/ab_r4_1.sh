# round 4, first GPU call: parity of the unified front end (am_k_fe4<32,1,3>) + A/B against round 3's am_k_fe3
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" "FE=3 LIB=build/var/lib_nw2.so" "FE=3 LIB=build/var/lib_nw6.so" "FE=3 LIB=build/var/lib_abl1.so" 2>&1 | tee gpurun_out/ab_r4_1.txt
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_r3.so" 2>&1 | tee -a gpurun_out/ab_r4_1.txt
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe4prof.so timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "^fe4" | tail -4 | tee gpurun_out/fe4_phase_clocks.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-extra > gpurun_out/bench_r4_1.json 2> gpurun_out/bench_r4_1.err; tail -c 600 gpurun_out/bench_r4_1.json
