#!/bin/bash
# round 5 re-entry: every GPU test, then per-op table (one stream) and the headline line
set -o pipefail
OUT=gpurun_out/${OUTDIR:-r05s}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -30 $OUT/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
rc=$?; tail -3 $OUT/pytest_gpu.log; [ $rc -ne 0 ] && { grep -E "FAILED|Error|assert" $OUT/pytest_gpu.log | tail -20; exit $rc; }
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams 1 --per-op $OUT/per_op.json > $OUT/bench1.json 2> $OUT/bench1.err || { tail -5 $OUT/bench1.err; exit 1; }
timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams 2 > $OUT/bench2.json 2> $OUT/bench2.err || { tail -5 $OUT/bench2.err; exit 1; }
python - $OUT <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/per_op.json"))
b1 = json.load(open(sys.argv[1] + "/bench1.json")); b2 = json.load(open(sys.argv[1] + "/bench2.json"))
print(f"one stream {b1['ms_per_step']:.4f} ms/step {b1['value']:.0f} img/s | two streams {b2['ms_per_step']:.4f} ms/step {b2['value']:.0f} img/s")
print(" | ".join(f"{k.split(' ')[0]} {x['us']:.0f}us" for k, x in b1["roofline"]["families"].items()))
for o in d["ops"]:
    print(f"{o['us_net']:8.2f}  {o['op']}")
PY
