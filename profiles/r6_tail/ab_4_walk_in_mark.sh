# Round 6: the block walk by the marking workgroups themselves (group exits by am_k_cblk_exit's last-arriving blocks; default)
# against the one-workgroup am_k_cblk_walk launch (knobs build, AIRMODES_WALK_IN_MARK=0); device suite first
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for ARGS in "" "--lambda 2000"; do
for i in 1 2 3; do for f in 1 0; do AIRMODES_WALK_IN_MARK=$f AIRMODES_HIP_LIB=$K python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('walk_in_mark=$f $ARGS: ms/step %.4f  GS/s %.1f  fe_ms %.4f parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d.get('parity')))"; done; done; done
STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | head -9
