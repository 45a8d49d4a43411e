# Round 6, call 1: gather + rows + refinement in ONE launch with the rows in LDS (am_k_refine_seg, default) against round 5's
# am_k_gather_wg<1> + am_k_refine_late (test build, AIRMODES_FUSED_REFINE=0); the 64 Msps device tests first
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r6_1}
rm -rf $OUT; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/tests_gpu.txt
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"; }
run() { env AIRMODES_HIP_LIB=$K $2 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in 1 2 3; do
    run "fused    " AIRMODES_X=0
    run "r5 tail  " AIRMODES_FUSED_REFINE=0
  done
done
STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -10 $OUT/kstats.txt; head -10 $OUT/kstats_lambda2000.txt
