# pipelined throughput against the number of am_k_fe3 workgroups per CU (knobs build):  gpurun -- 'bash tools/ab_r3_7.sh'
export AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
for rep in 1 2; do
for w in 6 5 4 3; do
  AIRMODES_FE3_WGS_PER_CU=$w timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab7_$w.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab7_$w.json").read().strip().splitlines()[-1])
print("wgs/cu $w: serial %.1f GS/s  fe %.4f ms  pipelined %.1f GS/s  parity %s" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["pipelined"]["value"]/1e9, d.get("parity")))
PY
done; done
