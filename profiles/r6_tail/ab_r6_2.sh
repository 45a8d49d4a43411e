# Round 6: am_k_refine_seg's phase clocks (profiling build) + interleaved A/B of the fused refinement against round 5's two launches
export AIRMODES_FUSED_REFINE=1
P=$PWD/build/var/lib_rsprof.so
[ -f $P ] && { AIRMODES_HIP_LIB=$P timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra 2>&1 | grep rseg | tail -4; AIRMODES_HIP_LIB=$P timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra --lambda 2000 2>&1 | grep rseg | tail -4; }
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
for ARGS in "" "--lambda 2000"; do
for i in 1 2; do for f in 1 0; do AIRMODES_FUSED_REFINE=$f AIRMODES_HIP_LIB=$K python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused=$f $ARGS: ms/step %.4f  GS/s %.1f  fe_ms %.4f parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d.get('parity')))"; done; done; done
AIRMODES_HIP_LIB=$K STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | head -9
