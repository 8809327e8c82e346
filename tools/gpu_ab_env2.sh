#!/bin/bash
# same-visit A/B of an environment switch, two batches in flight AND one stream: tools/gpu_ab_env2.sh VAR "0 1" [bench args]
set -o pipefail
VAR=$1; VALS=$2; shift 2
mkdir -p gpurun_out/ab
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/ab/build.log 2>&1 || { tail -30 gpurun_out/ab/build.log; exit 1; }
for rep in 1 2; do
  for v in $VALS; do
    for st in 2 1; do
      env $VAR=$v timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams $st "$@" > gpurun_out/ab/${VAR}_${v}_${st}_${rep}.json 2> gpurun_out/ab/${VAR}_${v}_${st}_${rep}.err || { tail -5 gpurun_out/ab/${VAR}_${v}_${st}_${rep}.err; exit 1; }
      python - "$VAR" "$v" "$rep" "$st" <<'PY'
import json, sys
var, v, rep, st = sys.argv[1:5]
d = json.load(open(f"gpurun_out/ab/{var}_{v}_{st}_{rep}.json"))
f = d["roofline"]["families"]
print(f"{var}={v} streams {st} run {rep}: {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s launches {d['config']['launches_per_step']} | " +
      " | ".join(f"{k.split(' ')[0]} {x['us']:.0f}us" for k, x in f.items()), flush=True)
PY
    done
  done
done
