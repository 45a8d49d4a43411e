# am_k_fe4: reference-level rows parked in dead ring rows (wave 0) / the whole 8-row buffer (wave 1): parity suite, old / new interleaved
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4m_tests.txt
for w in 20msps 2msps; do
BENCH_ARGS="--workload $w" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_pre_f4park.so" "FE=3 LIB=default" 2>&1 | grep "^FE="
done > gpurun_out/r4m_ab.txt
cat gpurun_out/r4m_tests.txt gpurun_out/r4m_ab.txt
