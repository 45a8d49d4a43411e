# instruction scheduling strategies for the whole library (am_k_fe3 is what is watched):  gpurun -- 'bash tools/ab_r3_18.sh'
for rep in 1 2; do
for v in "" s_max-ilp s_max-memory-clause s_default; do
  if [ -n "$v" ]; then export AIRMODES_HIP_LIB=$PWD/build/var/lib_$v.so; else unset AIRMODES_HIP_LIB; fi
  timeout 200 python bench.py --no-cpu-baseline --no-extra > gpurun_out/ab18.json 2>gpurun_out/ab18.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab18.json").read().strip().splitlines()[-1])
    print("variant '%s': %.1f GS/s  %.4f ms/step  fe %.4f ms  frac %.3f parity %s" % ("$v", d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity")))
except Exception as e:
    print("variant '$v' failed", e)
PY
done; done
