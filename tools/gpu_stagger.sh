# start-up stagger sweep of the fused front-end kernel (tuning aid): WL=64msps STAGS="0 30000 45000"
for g in ${STAGS:-0 30000 45000 60000}; do
  AIRMODES_FE2_STAGGER=$g timeout 300 python bench.py --workload ${WL:-64msps} --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${WL:-64msps} stagger $g: fe_ms %.4f  ms/step %.3f'%(d['roofline']['kernel_ms'],d['ms_per_step']))"
done
