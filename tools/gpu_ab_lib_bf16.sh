#!/bin/bash
# In-visit A/B of builds of the library: rsgnet_amd/lib/<name>/librsgnet_hip.so for every name in LIBS (built from
# another commit or with other -D switches on the CPU box) against the in-tree one, interleaved, 3 rounds.
# Box-to-box variance is ~10 %, so only this kind of A/B counts.
mkdir -p gpurun_out
W=${WORKLOADS:-hrnet_w32_256x192}
LIBS=${LIBS:-base}
one() { python bench.py --workload $1 --steps 30 --warmup 8 --no-cpu-baseline --dtype bf16 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; }
for w in $W; do
  for round in 1 2 3; do
    for l in $LIBS; do echo -n "$w $l: "; RSGNET_HIP_LIB=$PWD/rsgnet_amd/lib/$l/librsgnet_hip.so one $w; echo; done
    echo -n "$w new: "; one $w; echo
  done
done
