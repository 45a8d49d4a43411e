# the round's last closing run (after the wave priority by step): smoke(), the plain bench line, the three rocprofv3 passes of the
# 64 Msps workload, the profiling build's timeline, the 20 / 2 Msps lines, a 2 000-step leg -- most important first
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/final2; rm -rf $OUT; mkdir -p $OUT
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err
STEPS=10 timeout 200 bash tools/gpu_prof.sh > $OUT/summary.txt 2>&1
cp gpurun_out/prof_stats/bench_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
cp gpurun_out/bench_prof.json $OUT/bench_under_rocprof.json 2>/dev/null
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof.so timeout 60 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-parity 2>&1 >/dev/null | grep "^fe3" | tail -32 > $OUT/fe3_timeline.txt 2>&1
timeout 100 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline --no-extra --no-parity > $OUT/bench_2000_steps.json 2>/dev/null
timeout 150 python bench.py --workload 20msps --no-cpu-baseline > $OUT/bench_20msps.json 2>/dev/null
timeout 150 python bench.py --workload 2msps --no-cpu-baseline > $OUT/bench_2msps.json 2>/dev/null
tail -2 $OUT/smoke.txt
python - <<'PY'
import json
for f in ['bench','bench_2000_steps','bench_20msps','bench_2msps']:
    try:
        d=json.load(open('gpurun_out/final2/%s.json'%f)); r=d['roofline']
    except Exception as e:
        print(f,'missing',e); continue
    print(f, 'GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f traffic %s parity %s'%(d['value']/1e9,d['ms_per_step'],r['kernel_ms'],r['frac'],r['traffic'],d.get('parity')))
    x=d.get('realistic_density')
    if x: print('   2000 bursts/s: GS/s %.1f ms/step %.4f kernel_ms %.4f frac %.3f parity %s'%(x['value']/1e9,x['ms_per_step'],x['kernel_ms'],x['roofline_frac'],x.get('parity')))
    x=d.get('pipelined')
    if x: print('   pipelined: GS/s %.1f ms/step %.4f'%(x['value']/1e9,x['ms_per_step']))
    x=d.get('k_streams_per_scan')
    if x: print('   k=8: GS/s %.1f ms/step %.4f frac %.3f; two in flight GS/s %.1f'%(x['value']/1e9,x['ms_per_step'],x['roofline_frac'],(x.get('two_scans_in_flight') or {}).get('value',0)/1e9))
PY
head -14 $OUT/summary.txt | tail -4; grep "am_k_fe3" $OUT/summary.txt | head -4; head -22 $OUT/fe3_timeline.txt | tail -16
