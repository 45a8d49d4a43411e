mkdir -p gpurun_out
for rep in 1 2; do
for v in default abl8 abl2 abl10 abl1 abl11; do
  if [ $v = default ]; then unset AIRMODES_HIP_LIB; else export AIRMODES_HIP_LIB=$PWD/build/var/lib_$v.so; fi
  timeout 150 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$v: ms/step %.3f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))
except Exception as e: print('$v: failed', e)"
done
done 2>&1 | tee gpurun_out/ab1.txt
