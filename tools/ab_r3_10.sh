# the other rates after a change to am_k_fe4:   gpurun -- 'bash tools/ab_r3_10.sh'
timeout 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "production_stages or sharded or streaming" 2>&1 | tail -2
for rep in 1 2; do
for wl in 20msps 2msps; do
  timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-extra > gpurun_out/ab10.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab10.json").read().strip().splitlines()[-1])
print("$wl: %.1f GS/s  %.4f ms/step  fe %.4f ms  frac %.3f  parity %s" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity")))
PY
done; done
