# pipelined throughput with one hardware queue per context (GPU_MAX_HW_QUEUES=8) against am_k_fe3's workgroups per CU:  gpurun -- 'bash tools/ab_r3_9.sh'
export AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
export GPU_MAX_HW_QUEUES=8
for rep in 1 2; do
for w in 6 5 4; do
  AIRMODES_FE3_WGS_PER_CU=$w timeout 200 python bench.py --no-cpu-baseline > gpurun_out/ab9_$w.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab9_$w.json").read().strip().splitlines()[-1])
print("wgs/cu $w: serial %.1f GS/s  fe %.4f ms  pipelined %.1f GS/s  realistic %.1f" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["pipelined"]["value"]/1e9, d["realistic_density"]["value"]/1e9))
PY
done; done
