# Round 5, call 17: wave priority by step (fes_step_priority, now in am_k_fe3 and am_k_fe4) against the build without it
# (-DFES_STEP_PRIO=0), interleaved, all workloads + eight streams per scan; the device suite on the new default
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_17}
rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step'],d.get('parity')))"; }
run() { if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $PAR $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000" "--workload 20msps" "--workload 2msps" "--workload 20msps --streams 8" "--workload 2msps --streams 8"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  case "$ARGS" in *streams*) PAR="--no-parity"; REPS="1";; *) PAR=""; REPS="1 2";; esac
  for rep in $REPS; do
    run "priority" default
    run "without " $PWD/build/var/lib_noprio.so
  done
done
cat $OUT/tests_gpu.txt $OUT/ab.txt
