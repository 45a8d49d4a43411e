# A/B wall-clock comparison of two environment settings, interleaved (bench noise is ~5 %)
# usage: A="VAR=1" B="VAR=0" [PIPE=1] bash tools/gpu_ab.sh
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    env $cfg python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${PIPE:+--inflight 3} --no-pipelined 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg: ms/step %.3f  GS/s %.1f  fe %.3f'%(d['ms_per_step'],d['value']/1e9,d['roofline']['kernel_ms']))"
  done
done
