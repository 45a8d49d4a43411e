#!/bin/bash
# GPU box: micro-benchmarks behind the front-end redesign (results -> gpurun_out/ubench/)
mkdir -p gpurun_out/ubench
cd tools/ubench
for f in "$@"; do
  echo "== $f" 
  timeout 300 ./$f 2>&1 | tee ../../gpurun_out/ubench/$f.txt
done
