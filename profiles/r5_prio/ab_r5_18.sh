# Round 5, call 18: wave priority by turn in the extraction kernel's grid-stride loop against the build without it (-DAM_XS_PRIO=0)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_18}
rm -rf $OUT; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d parity %s'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step'],d.get('parity')))"; }
run() { if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000"; do if [ -n "$ARGS" ]; then REPS="1"; else REPS="1 2"; fi
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in $REPS; do
    run "xs priority" default
    run "without    " $PWD/build/var/lib_noxsprio.so
  done
done
STEPS=10 timeout 100 bash tools/gpu_kstats.sh 2>&1 | grep -E "extract_slice|gather_wg|refine_late" > $OUT/kstats.txt
AIRMODES_HIP_LIB=$PWD/build/var/lib_noxsprio.so STEPS=10 timeout 100 bash tools/gpu_kstats.sh 2>&1 | grep -E "extract_slice|gather_wg|refine_late" > $OUT/kstats_without.txt
cat $OUT/ab.txt; echo with; cat $OUT/kstats.txt; echo without; cat $OUT/kstats_without.txt
