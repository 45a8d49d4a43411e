# K independent streams per scan (am_process_multi) on the device: the parity suite, then the 2 / 20 Msps workloads with 1, 4, 8
# and 16 receivers' seconds in one scan, and the default 64 Msps line as the box's yardstick
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kstreams; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-extra > $OUT/bench_64msps.json 2> $OUT/bench.err
for W in 20msps 2msps; do
  timeout 400 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_${W}.json 2>> $OUT/bench.err
  for K in 4 8 16; do
    timeout 400 python bench.py --workload $W --streams $K --no-cpu-baseline --no-extra > $OUT/bench_${W}_k$K.json 2>> $OUT/bench.err
  done
done
cat $OUT/tests_gpu.txt; tail -1 $OUT/smoke.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/kstreams/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], 'GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f path %.3f parity %s'%(d['value']/1e9,d['ms_per_step'],r['kernel_ms'],r['frac'],r['path_frac_of_hbm_peak'],d.get('parity')), d.get('k_streams',{}).get('parity_every_stream'))
    x=d.get('k_streams_per_scan')
    if x: print('   extra k=8: GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f parity %s'%(x['value']/1e9,x['ms_per_step'],x['kernel_ms'],x['roofline_frac'],x.get('parity_every_stream')))
PY
tail -5 $OUT/bench.err
