# quick GPU loop: parity tests, bench lines, kernel stats
mkdir -p gpurun_out
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
fi
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
python -c "
import json;d=json.load(open('gpurun_out/bench.json'));print('64msps value %.2f GS/s ms/step %.3f fe_ms %.3f frac %.3f parity %s cpu %.1f MS/s'%(d['value']/1e9,d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['frac'],d.get('parity'),d['cpu_baseline']['value']/1e6))"
tail -1 gpurun_out/bench.err
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_prof.json 2> $OUT/bench_prof.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py gpurun_out | head -18
