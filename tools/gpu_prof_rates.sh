# the three rocprofv3 passes (stats, FETCH_SIZE, WRITE_SIZE) for the 20 and 2 Msps workloads:  gpurun -- 'bash tools/gpu_prof_rates.sh'
mkdir -p gpurun_out/rates
for w in 20msps 2msps; do
  BENCH_ARGS="--workload $w" STEPS=10 timeout 300 bash tools/gpu_prof.sh > gpurun_out/rates/summary_$w.txt 2>&1
  grep -E "fe4" gpurun_out/rates/summary_$w.txt | head -4
done
