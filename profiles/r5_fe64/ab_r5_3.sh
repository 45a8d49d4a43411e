# Round 5, call 3 (gpurun -- 'bash tools/ab_r5_3.sh'): device tests, then interleaved on one box
#   default (dense rows from IQ in am_k_gather_wg, 5 waves/SIMD; extraction loads a long packet's second half early)
#   wps4    the gather + rows kernel compiled for 4 waves per SIMD (no spills)
#   early   am_k_fe3 issues the next step's loads under its sparse outputs (FE3_EARLY=1)
#   cached  am_k_fe3 loads the samples with the default cache policy instead of nt
#   xtail0  extraction: chips 128.. loaded behind the long / short decision (round 4)
#   rowsfe  rows by the front end (round 4 arrangement; knobs build + AIRMODES_ROWS_FE=1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_3}
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
    run "wps4   " $PWD/build/var/lib_wps4.so
    run "early  " $PWD/build/var/lib_early.so
    run "cached " $PWD/build/var/lib_cached.so
    run "xtail0 " $PWD/build/var/lib_xtail0.so
    run "rowsfe " $K AIRMODES_ROWS_FE=1
  done
done
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
for v in early cached xtail0 wps4; do AIRMODES_HIP_LIB=$PWD/build/var/lib_$v.so STEPS=10 timeout 200 bash tools/gpu_kstats.sh > $OUT/kstats_$v.txt 2>&1; done
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -9 $OUT/kstats.txt
