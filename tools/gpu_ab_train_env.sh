#!/bin/bash
# same-visit A/B of an environment switch on the training step: tools/gpu_ab_train_env.sh VAR "0 1" [bench args]; two runs per
# value, interleaved
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
for rep in 1 2; do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/ab/train_${VAR}_${v}_${rep}.json 2> gpurun_out/ab/train_${VAR}_${v}_${rep}.err || { tail -5 gpurun_out/ab/train_${VAR}_${v}_${rep}.err; exit 1; }
    echo "$VAR=$v run $rep: $(tail -1 gpurun_out/ab/train_${VAR}_${v}_${rep}.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); f=d["roofline"]["families"]; print(d["ms_per_step"], "ms/step", d["value"], "img/s |", " | ".join(k+" %.0fus" % v["us"] for k,v in f.items()), "| loss", d["config"]["loss_first_last"])')"
  done
done
