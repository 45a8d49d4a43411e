# Round 6: the dominant kernel's start / stop times through hipExtLaunchKernelGGL (no event packets around the launch; build/var/lib_extev.so)
# against two hipEventRecord packets around it (HEAD before the change; build/var/lib_head.so), interleaved
for ARGS in "" "--lambda 2000"; do for i in 1 2 3 4; do for L in extev head; do AIRMODES_HIP_LIB=$PWD/build/var/lib_$L.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra --no-parity $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L $ARGS: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac']))"; done; done; done
