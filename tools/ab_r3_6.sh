mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('direct : ms/step %.3f  GS/s %.1f  fe_ms %.4f'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms']))"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --force-sharded 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('sharded: ms/step %.3f  GS/s %.1f  parity %s'%(d['ms_per_step'],d['value']/1e9,d.get('parity')))"
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra --seconds 0.25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('0.25 s : ms/step %.3f  GS/s %.1f  fe_ms %.4f'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms']))"
STEPS=10 BENCH_ARGS="--seconds 0.25" bash tools/gpu_kstats.sh 2>&1 | head -12
} 2>&1 | tee gpurun_out/ab6.txt
