for x in 0 2048 4096 20000 60000; do
  AIRMODES_FE2_LDS_EXTRA=$x timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/occ_$x.json 2>/dev/null
  python -c "
import json;d=json.load(open('gpurun_out/occ_$x.json'));print('lds extra $x: fe_ms %.3f  ms/step %.3f'%(d['roofline']['kernel_ms'],d['ms_per_step']))"
done
