# after a change to the tail kernels: parity, per-kernel times at 64 and 2 Msps, step times:  gpurun -- 'bash tools/ab_r3_14.sh'
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
bash tools/gpu_kstats.sh 2>&1 | grep -E "am_k|ms/step"
BENCH_ARGS="--workload 2msps" bash tools/gpu_kstats.sh 2>&1 | grep -E "cblk|extract|ms/step"
for rep in 1 2; do for wl in 64msps 20msps 2msps; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-extra > gpurun_out/ab14.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab14.json").read().strip().splitlines()[-1])
print("$wl: %.1f GS/s  %.4f ms/step parity %s" % (d["value"]/1e9, d["ms_per_step"], d.get("parity")))
PY
done; done
