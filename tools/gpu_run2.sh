mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
# order check: library first, torch afterwards, in one process
python - > gpurun_out/order.log 2>&1 <<'PY'
import sys; sys.path.insert(0,'gr-air-modes_amd')
import air_modes
c=air_modes.Context(4e6)
import torch
print("torch sees", torch.cuda.device_count(), "devices after our library was loaded first")
x=torch.ones(4,device='cuda'); print(float(x.sum()))
PY
cat gpurun_out/order.log | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/bench_prof.err
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/prof_r1 | head -20
python - <<'PY'
import glob,csv
for f in glob.glob('gpurun_out/prof_r1/**/*kernel_stats.csv', recursive=True):
    print(f)
    for i,row in enumerate(csv.reader(open(f))):
        if i<25: print(row[:8])
PY
