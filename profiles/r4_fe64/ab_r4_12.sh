# reference-level rows parked in dead ring rows (wave 0) / eight buffer rows (wave 1): parity suite, old / new interleaved, clocks
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4j_tests.txt
bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_pre_avg.so" "FE=3 LIB=default" > gpurun_out/r4j_ab.txt 2>&1
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_pre_avg.so" "FE=3 LIB=default" >> gpurun_out/r4j_ab.txt 2>&1
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra 2> gpurun_out/r4j_clocks.txt >/dev/null
cat gpurun_out/r4j_tests.txt gpurun_out/r4j_ab.txt; grep "fe3 clocks" gpurun_out/r4j_clocks.txt | head -2
