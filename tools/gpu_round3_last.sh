#!/bin/bash
# end of round 3: the driver's command on the final build (two batches in flight), rocprofv3 of the one-stream form of it (settled per-kernel durations), the other configs
# command, + the other configs; every line carries config.cold_start = the region timed the way rounds 1-2 timed it
mkdir -p gpurun_out/r03p; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
O=gpurun_out/r03p
one() { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.log 2>&1 || { tail -5 $O/$n.log; exit 1; }; tail -1 $O/$n.log > $O/$n.json; python - $O/$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d["dtype"], "launches", d["config"].get("launches_per_step"),
      "cold", (d["config"].get("cold_start") or {}).get("ms_per_step"), "whole frac", (r.get("whole_step") or {}).get("frac", r.get("frac")),
      "dominant frac", r.get("frac"), "other", (d.get("other_dtype") or {}).get("ms_per_step"), flush=True)
PY
}
one bench --gpus 1 --steps 20 --warmup 5 --per-op $O/per_op.json || exit 1
rm -rf $O/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --streams 1 > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/trace_steady.py $O/prof > $O/trace_steady.txt 2>&1; head -40 $O/trace_steady.txt
find $O/prof -name "*.csv" -size +1M -delete
one bf16_w32 --workload hrnet_w32_256x192 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
one w48 --workload hrnet_w48_384x288 --steps 20 --warmup 5 --no-cpu-baseline
one bf16_w48 --workload hrnet_w48_384x288 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
one rsg --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline
one rsg48 --workload rsgnet_w48_384x288_coco --steps 10 --warmup 3 --no-cpu-baseline
one cp --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline
one train_hrnet --train --steps 10 --warmup 3 --no-cpu-baseline
one train_rsgnet --train --workload rsgnet_w32_256x192_coco --steps 10 --warmup 3 --no-cpu-baseline
echo visit done
