# Round 6: bitmap words per am_k_refine_seg workgroup (shares of a 1 344-word segment): 448 (3 shares), 560 (3 shares, uneven), 672 (2, default)
for v in rs448 rs560 default; do
  L=$PWD/build/var/lib_$v.so; [ $v = default ] && L=$PWD/gr-air-modes_amd/csrc/libairmodes_hip.so
  for ARGS in "" "--lambda 2000"; do
    AIRMODES_HIP_LIB=$L BENCH_ARGS="$ARGS" STEPS=10 bash tools/gpu_kstats.sh 2>/dev/null | grep -E "refine_seg|ms/step" | tr '\n' ' '; echo " <- $v $ARGS"
  done
done
