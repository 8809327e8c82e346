#!/bin/bash
# per-op table (one stream) for each RSGNET_HIP_TUNE value in VALS; prints the first OPS ops
mkdir -p gpurun_out/r05f
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05f/build.log 2>&1 || { tail -30 gpurun_out/r05f/build.log; exit 1; }
for rep in 1 2; do for v in $VALS; do
  RSGNET_HIP_TUNE=$v timeout -k 10 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --streams 1 --per-op gpurun_out/r05f/per_op_$v.json > gpurun_out/r05f/b_$v.json 2> gpurun_out/r05f/b_$v.err || { tail -5 gpurun_out/r05f/b_$v.err; exit 1; }
  python - gpurun_out/r05f/per_op_$v.json $v ${OPS:-8} <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], " | ".join(f"{o['op'][:28]} {o['us_net']:.1f}" for o in d["ops"][:int(sys.argv[3])]), flush=True)
PY
done; done
