export AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
for rep in 1 2; do
for w in 0 6 7; do
  AIRMODES_FE4_WGS_PER_CU=$w timeout 200 python bench.py --workload 20msps --no-cpu-baseline --no-extra > gpurun_out/ab11.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab11.json").read().strip().splitlines()[-1])
print("20msps wgs/cu $w: %.1f GS/s  %.4f ms/step  fe %.4f ms  frac %.3f" % (d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done; done
