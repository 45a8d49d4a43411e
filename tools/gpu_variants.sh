# compare tuning builds of the library (build/var/*.so, made by hand with -DFE2_NT=.. -DFE2_WPS=..)
mkdir -p gpurun_out
for lib in default build/var/*.so; do
  if [ "$lib" = default ]; then unset AIRMODES_HIP_LIB; else export AIRMODES_HIP_LIB=$PWD/$lib; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/var.json 2>gpurun_out/var.err
  python -c "
import json;d=json.load(open('gpurun_out/var.json'));print('$lib: fe_ms %.3f  ms/step %.3f  GS/s %.1f  pipelined %.1f'%(d['roofline']['kernel_ms'],d['ms_per_step'],d['value']/1e9,d['pipelined']['value']/1e9))" || tail -3 gpurun_out/var.err
done
