# extraction at 64 Msps with coalesced loads through LDS: parity suite, old / new interleaved (stress and realistic density), kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4o_tests.txt
bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_pre_xstage.so" "FE=3 LIB=default" > gpurun_out/r4o_ab.txt 2>&1
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_pre_xstage.so" "FE=3 LIB=default" 2>&1 | head -2 >> gpurun_out/r4o_ab.txt
STEPS=10 timeout 100 bash tools/gpu_kstats.sh > gpurun_out/r4o_kstats.txt 2>&1
cat gpurun_out/r4o_tests.txt gpurun_out/r4o_ab.txt; grep calls gpurun_out/r4o_kstats.txt | head -3
