# the device suite on the final sources, then the plain 20 / 2 Msps lines (whose k_streams_per_scan object carries the eight-stream
# figure and the two-scans-in-flight-from-one-thread figure), and 64 Msps with two streams per scan
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/kpipe; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/tests_gpu.txt
cat $OUT/tests_gpu.txt
for W in 20msps 2msps; do
  timeout 400 python bench.py --workload $W --no-cpu-baseline > $OUT/bench_${W}.json 2>> $OUT/bench.err
done
timeout 300 python bench.py --streams 2 --no-cpu-baseline --no-extra > $OUT/bench_64msps_k2.json 2>> $OUT/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extra > $OUT/bench_64msps.json 2>> $OUT/bench.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/kpipe/bench_*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f.split('/')[-1], 'GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f path %.3f parity %s'%(d['value']/1e9,d['ms_per_step'],r['kernel_ms'],r['frac'],r['path_frac_of_hbm_peak'],d.get('parity')))
    x=d.get('k_streams_per_scan')
    if x:
        print('   k=8: GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f parity %s'%(x['value']/1e9,x['ms_per_step'],x['kernel_ms'],x['roofline_frac'],x.get('parity_every_stream')))
        y=x.get('two_scans_in_flight')
        if y: print('   k=8, two in flight, one thread: GS/s %.1f ms/step %.4f path %.3f same %s'%(y['value']/1e9,y['ms_per_step'],y['path_frac_of_hbm_peak'],y['same_packets_as_one_scan_at_a_time']))
PY
tail -3 $OUT/bench.err
