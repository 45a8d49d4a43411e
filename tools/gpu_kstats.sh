# per-kernel average durations of one bench configuration (kernel trace only):   gpurun -- 'bash tools/gpu_kstats.sh'
#   BENCH_ARGS="--lambda 2000" for the realistic density; prints the top kernels
OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-10} --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/kstats/**/bench_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows[:18]:
    print('%-60s calls %4s avg_us %8.1f pct %5s' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
d = json.loads(open('gpurun_out/kstats/bench.json').read().strip().splitlines()[-1])
print('ms/step %.3f  pk %d' % (d['ms_per_step'], d['packets_per_step']))
PY
