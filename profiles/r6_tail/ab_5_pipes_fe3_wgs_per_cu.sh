# Round 6: batches / chunks in flight with am_k_refine_seg behind am_k_fe3 -- how many persistent front-end workgroups per CU leave room
# (26 KB of LDS each; a refine_seg workgroup wants 36 KB): AIRMODES_FE3_WGS_PER_CU = 4 / 5 / 6 under both pipes (knobs build)
K=$PWD/tests/gpu_variants/libairmodes_hip_knobs.so
for rep in 1 2; do for w in 4 5 6; do AIRMODES_FE3_WGS_PER_CU=$w AIRMODES_HIP_LIB=$K python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-parity --sustained-steps 0 --stream-seconds 12 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fe3 wgs/cu $w: single %.4f ms | 4 batches in flight %.1f GS/s | stream, 4 chunks in flight %.1f GS/s' % (d['ms_per_step'], d['pipelined']['value']/1e9, d['pipelined_stream']['value']/1e9))"; done; done
