# rocprofv3 passes for bench.py (run on the GPU box through gpurun):
#   1. kernel trace + stats   2. PMC FETCH_SIZE   3. PMC WRITE_SIZE   (separate passes)
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=${STEPS:-5}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS:-} > $OUT/bench_prof.json 2> $OUT/bench_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra ${BENCH_ARGS:-} > $OUT/bench_fetch.json 2> $OUT/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra ${BENCH_ARGS:-} > $OUT/bench_write.json 2> $OUT/bench_write.err
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -type f | head -30
python tools/prof_summary.py gpurun_out
