#!/bin/bash
# same-visit A/B of an environment switch on the training step: tools/gpu_ab_train_env.sh VAR "1 0" [bench args]
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/ab/build.log 2>&1 || { tail -30 gpurun_out/ab/build.log; exit 1; }
for rep in 1 2; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline "$@" 2> gpurun_out/ab/train_${VAR}_${v}_${rep}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v run $rep:', d['ms_per_step'], 'ms/step', d['value'], 'img/s')" || { tail -5 gpurun_out/ab/train_${VAR}_${v}_${rep}.err; exit 1; }
  done
done
