# what the sparse outputs of am_k_fe4 cost at 20 and 2 Msps: ablation builds (results invalid) beside the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 20msps 2msps; do
BENCH_ARGS="--workload $w" bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_f4abl16.so" "FE=3 LIB=build/var/lib_f4abl32.so" "FE=3 LIB=build/var/lib_f4abl1.so" 2>&1 | grep "^FE=" | head -4
done > gpurun_out/r4l_abl.txt
cat gpurun_out/r4l_abl.txt
