# fe4 VALU diet: parity suite, then old / new interleaved at 20 and 2 Msps (and 64 Msps as a check)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4e_tests.txt
BENCH_ARGS="--workload 20msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_fe4old.so" "FE=3 LIB=default" > gpurun_out/r4e_ab.txt 2>&1
BENCH_ARGS="--workload 2msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_fe4old.so" "FE=3 LIB=default" >> gpurun_out/r4e_ab.txt 2>&1
bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_fe4old.so" "FE=3 LIB=default" >> gpurun_out/r4e_ab.txt 2>&1
cat gpurun_out/r4e_tests.txt gpurun_out/r4e_ab.txt
