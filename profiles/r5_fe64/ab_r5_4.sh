# Round 5, call 4: device tests on the slimmed product library; dense rows with the flag loads batched (default) against
#   wps6   the gather + rows kernel compiled for 6 waves per SIMD (80 VGPRs, spills: all 1 489 workgroups resident at once)
#   rowsfe rows by the front end (round 4 arrangement; knobs build + AIRMODES_ROWS_FE=1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_4}
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"; }
run() { # name lib [env]
  if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L $3 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt
}
for ARGS in "" "--lambda 2000"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in 1 2 3; do
    run "default" default
    run "wps6   " $PWD/build/var/lib_wps6.so
    run "rowsfe " $K AIRMODES_ROWS_FE=1
  done
done
STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
AIRMODES_HIP_LIB=$PWD/build/var/lib_wps6.so STEPS=10 timeout 200 bash tools/gpu_kstats.sh > $OUT/kstats_wps6.txt 2>&1
if [ -f build/var/lib_fe3prof.so ]; then
  AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "^fe3" | tail -3 > $OUT/fe3_phase_clocks.txt
fi
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -9 $OUT/kstats.txt; cat $OUT/fe3_phase_clocks.txt
