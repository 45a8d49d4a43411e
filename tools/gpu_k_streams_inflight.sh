# K streams per scan with 1 / 2 / 3 scans in flight (one context and one host thread each): how much of the eight-stream step is
# the host's (packet copies, sorting into streams) and hides behind the next scan's kernels
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kflight; rm -rf $OUT; mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
tail -3 $OUT/smoke.txt
for W in 20msps 2msps; do
  for F in 1 2 3; do
    timeout 300 python bench.py --workload $W --streams 8 --inflight $F --steps 24 --warmup 6 --no-cpu-baseline --no-extra > $OUT/bench_${W}_k8_f$F.json 2>> $OUT/bench.err
  done
done
timeout 300 python bench.py --replicas --streams 8 --no-cpu-baseline --no-extra > $OUT/bench_replicas1_k8.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/kflight/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], 'GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f path %.3f parity %s'%(d['value']/1e9,d['ms_per_step'],r['kernel_ms'],r['frac'],r['path_frac_of_hbm_peak'],d.get('parity')))
PY
tail -3 $OUT/bench.err
