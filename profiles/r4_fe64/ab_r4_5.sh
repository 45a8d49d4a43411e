# round 4, fifth GPU call: chain places by fetch-add, am_k_refine_late with two positions per lane
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
for rep in 1 2; do
bash tools/gpu_ab_libs.sh "FE=3 LIB=default" 2>&1 | head -1 | tee -a gpurun_out/ab_r4_5.txt
done
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" 2>&1 | tee -a gpurun_out/ab_r4_5.txt
BENCH_ARGS="--workload 20msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" 2>&1 | tee -a gpurun_out/ab_r4_5.txt
BENCH_ARGS="--workload 2msps" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" 2>&1 | tee -a gpurun_out/ab_r4_5.txt
STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | tee gpurun_out/kstats_r4_5.txt
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | tee gpurun_out/kstats_r4_5_l2000.txt
timeout 200 python bench.py --force-sharded --no-cpu-baseline --no-extra > gpurun_out/bench_force_sharded.json 2> gpurun_out/bench_force_sharded.err; tail -c 1200 gpurun_out/bench_force_sharded.json; tail -3 gpurun_out/bench_force_sharded.err
timeout 300 python bench.py --steps 12 --warmup 3 > gpurun_out/bench_r4_5.json 2> gpurun_out/bench_r4_5.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r4_5.json'))
print({k:d[k] for k in ('value','ms_per_step','parity')}, d['roofline']['kernel_ms'], d['roofline']['frac'], d.get('pipelined',{}).get('value'), d.get('realistic_density',{}).get('ms_per_step'), d.get('realistic_density',{}).get('parity'))
PY
