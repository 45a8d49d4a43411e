# round 4: host overhead of the streaming time-shard step on one GPU (--force-sharded against the direct path, interleaved)
mkdir -p gpurun_out
for rep in 1 2 3; do
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct : ms/step %.4f fe_ms %.4f'%(d['ms_per_step'],d['roofline']['kernel_ms']))" | tee -a gpurun_out/ab_r4_6.txt
timeout 200 python bench.py --force-sharded --steps 20 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('sharded: ms/step %.4f fe_ms %.4f sync %s parity %s'%(d['ms_per_step'],d['roofline']['kernel_ms'],d.get('sharded_sync_steps'),d.get('parity')))" | tee -a gpurun_out/ab_r4_6.txt
done
BENCH_ARGS="--force-sharded" STEPS=10 timeout 300 bash tools/gpu_kstats.sh 2>&1 | tee gpurun_out/kstats_r4_6_sharded.txt
