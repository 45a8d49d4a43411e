mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log
timeout 400 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?" >> gpurun_out/bench.err
cat gpurun_out/bench.json; tail -2 gpurun_out/bench.err
timeout 400 python bench.py --steps 10 --warmup 2 --workload 2msps > gpurun_out/bench_2msps.json 2>> gpurun_out/bench.err
cat gpurun_out/bench_2msps.json
bash tools/gpu_prof.sh
