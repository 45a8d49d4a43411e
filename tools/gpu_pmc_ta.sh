OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_ta -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra > $OUT/ta.json 2> $OUT/ta.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
from collections import defaultdict
for f in glob.glob('gpurun_out/prof_ta/**/*counter_collection.csv', recursive=True):
    acc=defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:34]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'am_k' in k: print(k.ljust(36), {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
tail -2 $OUT/ta.err | cut -c1-200
