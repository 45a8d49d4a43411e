# what the sparse outputs of am_k_fe3 cost: ablation builds (results invalid) beside the default, and a plain bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/gpu_ab_libs.sh "FE=3 LIB=default" "FE=3 LIB=build/var/lib_abl16.so" "FE=3 LIB=build/var/lib_abl32.so" "FE=3 LIB=build/var/lib_abl1.so" 2>&1 | grep -v "^Traceback\|Error\|^  " > gpurun_out/r4i_abl.txt
timeout 400 python bench.py > gpurun_out/r4i_bench.json 2> gpurun_out/r4i_bench.err
cat gpurun_out/r4i_abl.txt; tail -c 400 gpurun_out/r4i_bench.json
