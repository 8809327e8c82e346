#!/bin/bash
# per-op table of the headline workload (one stream) under the environment given on the command line: VAR=val ... 
set -o pipefail
OUT=gpurun_out/${OUTDIR:-r05b}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -30 $OUT/build.log; exit 1; }
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams 1 --per-op $OUT/per_op.json "$@" > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/per_op.json"))
b = json.load(open(sys.argv[1] + "/bench.json"))
print(f"{b['ms_per_step']:.4f} ms/step {b['value']:.0f} img/s")
for o in d["ops"]:
    print(f"{o['us_net']:8.2f}  {o['op']}")
PY
