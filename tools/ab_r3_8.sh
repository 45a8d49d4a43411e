# am_k_fe4 workgroups per CU at 20 / 2 Msps (knobs build), and the host-threaded in-flight figure:  gpurun -- 'bash tools/ab_r3_8.sh'
export AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
for rep in 1 2; do
for wl in 20msps 2msps; do
for w in 0 5 6 7 8; do
  AIRMODES_FE4_WGS_PER_CU=$w timeout 200 python bench.py --workload $wl --no-cpu-baseline --no-extra > gpurun_out/ab8.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab8.json").read().strip().splitlines()[-1])
print("$wl wgs/cu $w: %.1f GS/s  fe %.4f ms  frac %.3f" % (d["value"]/1e9, d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done; done; done
unset AIRMODES_HIP_LIB
timeout 200 python bench.py --no-cpu-baseline --no-extra --inflight 3 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight 3 (host threads): %.1f GS/s %.4f ms/step' % (d['value']/1e9, d['ms_per_step']))"
