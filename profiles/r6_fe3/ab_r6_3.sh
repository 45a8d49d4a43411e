# Round 6: levelled segments in am_k_fe3 (1 536 workgroups of 13 / 14 steps, default) against 1 489 of 14 (test build,
# AIRMODES_FUSED_REFINE=0 keeps round 5's tail AND the uniform segments)
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for ARGS in "" "--lambda 2000"; do
for i in 1 2 3; do for f in 1 0; do AIRMODES_FUSED_REFINE=$f AIRMODES_HIP_LIB=$K python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused+levelled=$f $ARGS: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d.get('parity')))"; done; done; done
