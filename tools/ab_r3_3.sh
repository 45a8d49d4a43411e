mkdir -p gpurun_out
{
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra 2>&1 | grep "fe3" | tail -4
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --lambda 2000 2>&1 | grep "fe3" | tail -4
for rep in 1 2; do
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tee gpurun_out/bench_r3_3_$rep.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default: ms/step %.3f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step'],d.get('parity'))); print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','ms_per_step','kernel_ms','frac')}) for k,v in d.items() if k in ('pipelined','realistic_density')})"
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
} 2>&1 | tee gpurun_out/ab3.txt
