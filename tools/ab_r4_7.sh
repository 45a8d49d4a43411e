cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r4b_tests.txt
bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_old.so" "FE=3 LIB=default" > gpurun_out/r4b_ab.txt 2>&1
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_old.so" "FE=3 LIB=default" >> gpurun_out/r4b_ab.txt 2>&1
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof_new.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra 2> gpurun_out/r4b_clocks.txt >/dev/null
cat gpurun_out/r4b_tests.txt gpurun_out/r4b_ab.txt; grep "fe3" gpurun_out/r4b_clocks.txt | head -4
