# Round 5, call 1 (gpurun -- 'bash tools/ab_r5_1.sh'): device tests on the new sources, then, interleaved on one box,
#   rows by the front end (round 4: AIRMODES_ROWS_FE=1 in the knobs build)  vs  rows from IQ in am_k_gather_wg (default),
#   at the stress density and at 2 000 bursts/s; the level-2 block scans ablated (FE3_ABLATE=8, results invalid) to price
#   a tree order; kernel stats of the new default.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_1}
rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/tests_gpu.txt
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"; }
for args in "" "--lambda 2000"; do
  echo "== bench args: $args" >> $OUT/ab.txt
  for rep in 1 2 3; do
    AIRMODES_HIP_LIB=$K AIRMODES_ROWS_FE=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $args 2>/dev/null | line "rows_by_front_end" >> $OUT/ab.txt
    AIRMODES_HIP_LIB=$K python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $args 2>/dev/null | line "rows_from_iq     " >> $OUT/ab.txt
    if [ -f build/var/lib_abl8.so ]; then
      AIRMODES_HIP_LIB=$PWD/build/var/lib_abl8.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $args 2>/dev/null | line "rows_from_iq+noscan(INVALID)" >> $OUT/ab.txt
    fi
  done
done
timeout 300 python bench.py > $OUT/bench.json 2> $OUT/bench.err
STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -12 $OUT/kstats.txt; tail -c 600 $OUT/bench.json
