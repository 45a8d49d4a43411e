# do the kernels of batches in flight overlap?  kernel trace of the pipelined loop:   gpurun -- 'bash tools/gpu_pipe_trace.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out/pipetrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 9 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/pipetrace/**/bench_kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name'].replace('void ', '')[:28]) for r in rows]
ev.sort()
queues = sorted(set(e[2] for e in ev))
print('queues seen:', queues, ' dispatches:', len(ev))
# the last 80 dispatches (the pipelined loop runs last... the realistic-density figure follows it: take the fe3 launches of 3 queues)
multi = [e for e in ev if e[3].startswith('am_k_')]
# find the window where three different queues launch am_k_fe3
fe = [e for e in multi if e[3].startswith('am_k_fe3')]
qs = {}
for e in fe: qs.setdefault(e[2], []).append(e)
print({q: len(v) for q, v in qs.items()})
pipeq = [q for q, v in qs.items()]
t0 = None
# window: from the first fe3 of the second-most-used queue to its last
if len(pipeq) >= 2:
    second = sorted(qs.items(), key=lambda kv: len(kv[1]))[0][0]
    lo, hi = qs[second][2][0], qs[second][-1][1]
    win = [e for e in multi if e[0] >= lo and e[1] <= hi]
    busy_any = 0; busy_sum = 0
    pts = sorted(set([e[0] for e in win] + [e[1] for e in win]))
    for a, b in zip(pts[:-1], pts[1:]):
        k = sum(1 for e in win if e[0] <= a and e[1] >= b)
        if k: busy_any += b - a
        busy_sum += k * (b - a)
    print('window %.1f us: some kernel running %.1f us, sum of kernel durations %.1f us (overlap factor %.2f)' % ((hi - lo) / 1e3, busy_any / 1e3, busy_sum / 1e3, busy_sum / max(busy_any, 1)))
    nfe = sum(1 for e in win if e[3].startswith('am_k_fe3'))
    print('fe3 launches in window:', nfe, ' -> %.1f us per batch' % ((hi - lo) / 1e3 / max(nfe, 1)))
    alone = sum(min(e[1], hi) - max(e[0], lo) for e in win if e[3].startswith('am_k_fe3'))
    print('am_k_fe3 running: %.1f us of the window' % (alone / 1e3))
    for e in win[:75]:
        print('%9.1f %9.1f q%d %s' % ((e[0] - lo) / 1e3, (e[1] - lo) / 1e3, e[2], e[3]))
PY
