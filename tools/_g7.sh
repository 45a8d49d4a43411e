cd /root/repo
for i in 1 2 3; do
timeout 400 python bench.py --steps 200 --force-sharded --backend nccl --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g7_nccl_inflight_$i.json"
timeout 400 python bench.py --steps 200 --force-sharded --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g7_nogroup_inflight_$i.json"
done
