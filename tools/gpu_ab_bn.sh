#!/bin/bash
# In-visit A/B: base vs new library, Bottleneck fusion off/on
mkdir -p gpurun_out
run() { for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; done; echo; }
for bn in 0 1; do
  export RSGNET_HIP_PLAN="bottleneck=$bn"
  echo "== base bn=$bn"; RSGNET_HIP_LIB=$PWD/rsgnet_amd/lib/base/librsgnet_hip.so run
  echo "== new bn=$bn";  run
done
