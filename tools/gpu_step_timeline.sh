# kernel timeline of three steps under rocprofv3 --kernel-trace:   gpurun -- 'bash tools/gpu_step_timeline.sh'
# (default: the time-sharded step through a world-1 RCCL group; BENCH_ARGS="" = the plain single-GPU step; durations and gaps under the tracer are inflated)
OUT=$GRAFT_REPO_ROOT/gpurun_out/steptl
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-extra ${BENCH_ARGS---force-sharded --backend nccl} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/steptl/**/bench_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name'].replace('void ', '')[:60]) for r in rows)
fe = [i for i, e in enumerate(ev) if e[3].startswith('am_k_fe3')]
# the timed loop: 12 steps after 2 (+ in-flight warmup) -- show from the 6th front end launch on, three steps
i0 = fe[6]; i1 = fe[9]
t0 = ev[i0][0]
out = open('gpurun_out/steptl/timeline.txt', 'w')
for e in ev[i0:i1 + 1]:
    out.write('%9.1f %9.1f  dur %7.1f  q %d  %s\n' % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3]))
out.write('front end to front end: %s us\n' % [round((ev[b][0] - ev[a][0]) / 1e3, 1) for a, b in zip(fe[4:12], fe[5:13])])
PY
grep '^{' $OUT/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('sharded_steps_in_flight'))" >> $OUT/timeline.txt
