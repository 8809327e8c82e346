#!/bin/bash
# A/B of execution modes in one visit
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
for mode in "RSGNET_HIP_PLAN=fuse_module=0" "RSGNET_HIP_PLAN=fuse_module=1" "RSGNET_HIP_PLAN=fuse_module=0" "RSGNET_HIP_PLAN=fuse_module=1" ${EXTRA_MODES}; do
  echo "== $mode"
  env $mode timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, \|"ms_per_step": [0-9.]*' | tr '\n' ' '
  echo
done
