# the round's closing run on one box: the device parity suite, smoke(), then the plain bench lines (every one with parity and,
# once profiles/current_traffic.json is stamped for the sources, with traffic)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final; rm -rf $OUT; mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 200 python bench.py --workload 20msps --no-cpu-baseline --no-extra > $OUT/bench_20msps.json 2>/dev/null
timeout 200 python bench.py --workload 2msps --no-cpu-baseline --no-extra > $OUT/bench_2msps.json 2>/dev/null
timeout 200 python bench.py --force-sharded --no-cpu-baseline --no-extra > $OUT/bench_force_sharded.json 2>/dev/null
cat $OUT/tests_gpu.txt; tail -2 $OUT/smoke.txt
python - <<'PY'
import json
for f in ['bench','bench_20msps','bench_2msps','bench_force_sharded']:
    d=json.load(open('gpurun_out/final/%s.json'%f)); r=d['roofline']
    print(f, 'GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f traffic %s parity %s'%(d['value']/1e9,d['ms_per_step'],r['kernel_ms'],r['frac'],r['traffic'],d.get('parity')))
PY
# a long leg (2 000 steps = 0.6 s of device work) so that a sampler of GPU activity sees the device busy
timeout 200 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_2000_steps.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_2000_steps.json')); print('2000 steps: GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f'%(d['value']/1e9,d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac']))
PY
