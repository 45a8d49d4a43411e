# Round 5, call 16: wave priority rotating with the step in am_k_fe3 (tuning builds -DFE3_PRIO=1..4) against the default, interleaved;
# the profiling build's timeline with rotation 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_16}
rm -rf $OUT; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step'],d.get('parity')))"; }
run() { if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in 1 2; do
    run "default" default
    for v in 1 2 3 4; do run "prio$v  " $PWD/build/var/lib_prio$v.so; done
  done
done
AIRMODES_HIP_LIB=$PWD/build/var/lib_prio1prof.so timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity 2>&1 >/dev/null | grep "^fe3" | tail -32 > $OUT/fe3_timeline_prio1.txt 2>&1
cat $OUT/ab.txt; head -24 $OUT/fe3_timeline_prio1.txt
