# interleaved comparison of library builds / front ends on the GPU box
# usage: bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=2 LIB=default" "FE=3 LIB=build/var/lib_noslp.so" ...
for rep in 1 2; do
for cfg in "$@"; do
  eval "$cfg"
  if [ "$LIB" = default ]; then unset AIRMODES_HIP_LIB; else export AIRMODES_HIP_LIB=$PWD/$LIB; fi
  AIRMODES_FE=$FE python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-extra ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg: ms/step %.3f  GS/s %.1f  fe_ms %.4f frac %.3f pk %d'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms'],d['roofline']['frac'],d['packets_per_step']))"
done
done
