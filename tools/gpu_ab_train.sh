#!/bin/bash
# same-visit A/B of an environment switch on the training benches: tools/gpu_ab_train.sh VAR "v1 v2" [workloads]
VAR=$1; VALS=$2; W=${3:-"hrnet_w32_256x192 rsgnet_w32_256x192_coco"}
mkdir -p gpurun_out/ab
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/ab/build.log 2>&1 || { tail -30 gpurun_out/ab/build.log; exit 1; }
for w in $W; do
  for rep in 1 2; do
    for v in $VALS; do
      echo -n "$w $VAR=$v run $rep: "
      env $VAR=$v timeout -k 10 300 python bench.py --train --workload $w --steps 12 --warmup 4 2>gpurun_out/ab/err.log | grep -o '"value": [0-9.]*, "unit": "images/s"\|"ms_per_step": [0-9.]*' | tr '\n' ' ' || { tail -5 gpurun_out/ab/err.log; exit 1; }
      echo
    done
  done
done
