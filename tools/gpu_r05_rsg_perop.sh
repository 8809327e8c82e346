#!/bin/bash
# RSGNet-W32: bench line + per-op table (head ops), one visit
mkdir -p gpurun_out/r05h
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05h/build.log 2>&1 || { tail -30 gpurun_out/r05h/build.log; exit 1; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline --per-op gpurun_out/r05h/per_op_rsg.json > gpurun_out/r05h/bench_rsg.json 2> gpurun_out/r05h/bench_rsg.err || { tail -5 gpurun_out/r05h/bench_rsg.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05h/per_op_rsg.json")); b = json.load(open("gpurun_out/r05h/bench_rsg.json"))
print(b["value"], b["ms_per_step"], (b.get("single_stream") or {}).get("ms_per_step"))
print(" | ".join(f"{o['op'][:30]} {o['us_net']:.1f}" for o in d["ops"] if not any(k in o["op"] for k in ("stage", "layer1", "transition", "stem"))))
PY
done
