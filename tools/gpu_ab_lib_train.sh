#!/bin/bash
# In-visit A/B of library builds on the training bench (see tools/gpu_ab_lib.sh): LIBS="name ..." [WORKLOADS]
mkdir -p gpurun_out
W=${WORKLOADS:-hrnet_w32_256x192}
one() { python bench.py --train --workload $1 --steps 12 --warmup 4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '; }
for w in $W; do
  for round in 1 2; do
    for l in ${LIBS:-base}; do echo -n "$w $l: "; RSGNET_HIP_LIB=$PWD/rsgnet_amd/lib/$l/librsgnet_hip.so one $w; echo; done
    echo -n "$w new: "; one $w; echo
  done
done
