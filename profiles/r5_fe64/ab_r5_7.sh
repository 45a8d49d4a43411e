# Round 5, call 7: what a tree order of the 48 chip totals (level 2 of the canonical sums) would be worth in am_k_fe3 --
# a TIMING variant (-DFE3_TREE_TIMING: 6 DPP steps per direction instead of 47 rounds; another rounding order, results not canonical)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG:-r5_7}
rm -rf $OUT; mkdir -p $OUT
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1: ms/step %.4f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"; }
run() { if [ "$2" = default ]; then L=""; else L="AIRMODES_HIP_LIB=$2"; fi
  env $L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra $ARGS 2>/dev/null | line "$1" >> $OUT/ab.txt; }
for ARGS in "" "--lambda 2000"; do
  echo "== bench args: $ARGS" >> $OUT/ab.txt
  for rep in 1 2 3 4; do
    run "default   " default
    run "tree(time)" $PWD/build/var/lib_tree.so
  done
done
cat $OUT/ab.txt
