cd /root/repo
for st in 20 200; do
timeout 400 python bench.py --steps $st --force-sharded --backend nccl --steps-in-flight --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g6_nccl_inflight_$st.json"
timeout 400 python bench.py --steps $st --force-sharded --steps-in-flight --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g6_nogroup_inflight_$st.json"
timeout 400 python bench.py --steps $st --force-sharded --backend nccl --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g6_nccl_sync_$st.json"
timeout 400 python bench.py --steps $st --force-sharded --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' > "gpurun_out/g6_nogroup_sync_$st.json"
done
