# Round 5, call 15: the chain's successor without a search in memory -- among the group's own candidates in LDS for one that fails,
# by the whole wave (64 + 8 probes) for one that passes -- against the commit before (prev: a galloping search per lane); device tests first
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_15}
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"; }
run() { if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000" "--workload 20msps"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in 1 2 3; do
    run "default" default
    run "prev   " $PWD/build/var/lib_prev.so
  done
done
STEPS=10 timeout 200 bash tools/gpu_kstats.sh > $OUT/kstats.txt 2>&1
BENCH_ARGS="--lambda 2000" STEPS=10 timeout 200 bash tools/gpu_kstats.sh > $OUT/kstats_lambda2000.txt 2>&1
cat $OUT/tests_gpu.txt $OUT/ab.txt; head -9 $OUT/kstats.txt; grep refine $OUT/kstats_lambda2000.txt
