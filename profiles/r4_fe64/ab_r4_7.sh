cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r4d_tests.txt
bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_old.so" "FE=3 LIB=build/var/lib_noearly.so" "FE=3 LIB=default" > gpurun_out/r4d_ab.txt 2>&1
BENCH_ARGS="--lambda 2000" bash tools/gpu_ab_libs.sh "FE=3 LIB=build/var/lib_noearly.so" "FE=3 LIB=default" >> gpurun_out/r4d_ab.txt 2>&1
AIRMODES_HIP_LIB=$PWD/build/var/lib_fe3prof_new.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-extra 2> gpurun_out/r4d_clocks.txt >/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4d_pmc -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r4d_sq.txt 2>&1
import csv,glob,collections
for f in glob.glob('gpurun_out/r4d_pmc/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:24]; acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='SQ_INSTS_VALU': n[k]+=1
    for k in acc:
        if 'fe3' in k: print(k, n[k], {c: v/max(n[k],1) for c,v in acc[k].items()})
PY
cat gpurun_out/r4d_tests.txt gpurun_out/r4d_ab.txt gpurun_out/r4d_sq.txt; grep "fe3 clocks" gpurun_out/r4d_clocks.txt | head -2
