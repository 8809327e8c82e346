#!/bin/bash
# In-visit A/B of two builds of the library: rsgnet_amd/lib/base/librsgnet_hip.so (built from another commit
# on the CPU box) against the in-tree one.  Box-to-box variance is ~10 %, so only this kind of A/B counts.
mkdir -p gpurun_out
W=${WORKLOADS:-hrnet_w32_256x192}
run() { for i in 1 2 3; do python bench.py --workload $1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; done; echo; }
for w in $W; do
  echo "== $w base"; RSGNET_HIP_LIB=$PWD/rsgnet_amd/lib/base/librsgnet_hip.so run $w
  echo "== $w new";  run $w
done
