# Round 5, call 5: dense rows, straight-line loads, 6 waves per SIMD (default) against rows by the front end (round 4 arrangement);
# then the pipe (four batches in flight) with every context on its own share of the CUs (AIRMODES_PIPE_CU_PARTS, knobs build).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_5}
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
    run "rowsfe " $K AIRMODES_ROWS_FE=1
  done
done
STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 300 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
pipe() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipelined']; print('$1: pipe ms/batch %.4f GS/s %.1f path_frac %.3f same %s | single ms/step %.4f'%(p['ms_per_step'],p['value']/1e9,p.get('path_frac_of_hbm_peak',0),p['same_packets_last_batch'],d['ms_per_step']))"; }
for rep in 1 2; do
  for parts in 0 2 4; do
    AIRMODES_HIP_LIB=$K AIRMODES_PIPE_CU_PARTS=$parts timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity 2>/dev/null | pipe "cu_parts=$parts" >> $OUT/pipe_cu_parts.txt
  done
done
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -9 $OUT/kstats.txt; cat $OUT/pipe_cu_parts.txt
