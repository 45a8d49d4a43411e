# batches in flight (am_pipe, depth 4) with 6 / 5 / 4 fe3 workgroups per CU (knobs build: AIRMODES_FE3_WGS_PER_CU)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 6 5 4 6 5 4; do
  AIRMODES_HIP_LIB=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so AIRMODES_FE3_WGS_PER_CU=$w python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d.get('pipelined',{}); print('wgs/cu $w: ms/step %.3f fe_ms %.4f pipelined %.1f GS/s %.4f ms same %s'%(d['ms_per_step'], d['roofline']['kernel_ms'], p.get('value',0)/1e9, p.get('ms_per_step',0), p.get('same_packets_last_batch')))"
done > gpurun_out/r4g_pipe.txt 2>&1
cat gpurun_out/r4g_pipe.txt
