# SQ counters of the front-end kernel (two passes, kernel trace only):   gpurun -- 'bash tools/gpu_pmc.sh'
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/prof_pmc1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra ${BENCH_ARGS:-} > $OUT/pmc1.json 2> $OUT/pmc1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc2 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra ${BENCH_ARGS:-} > $OUT/pmc2.json 2> $OUT/pmc2.err
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
from collections import defaultdict
for d in ('prof_pmc1','prof_pmc2'):
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv'%d, recursive=True):
        acc=defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            import os
            if os.environ.get('KFILTER', 'am_k_fe') in k:
                print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
                r0=next(csv.DictReader(open(f)))
                print({kk:r0[kk] for kk in r0 if kk in ('VGPR_Count','Accum_VGPR_Count','SGPR_Count','LDS_Block_Size','Scratch_Size','Workgroup_Size','Grid_Size')})
PY
tail -3 $OUT/pmc1.err
