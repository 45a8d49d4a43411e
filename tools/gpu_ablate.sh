mkdir -p gpurun_out
for m in ${MASKS:-0 128 192 194 196 212 215}; do
  AIRMODES_FE2_ABLATE=$m timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$m.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/ab_$m.json'));print('ablate $m: fe_ms %.3f  ms/step %.3f'%(d['roofline']['kernel_ms'],d['ms_per_step']))"
done
