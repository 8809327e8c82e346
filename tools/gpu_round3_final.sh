#!/bin/bash
# the round's other-config numbers on the final build (no tests): bf16 lines, W48, RSGNet, CrowdPose flip test, training
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
one() { n=$1; shift; timeout -k 10 300 python bench.py "$@" > gpurun_out/$n.log 2>&1 || { tail -5 gpurun_out/$n.log; exit 1; }; tail -1 gpurun_out/$n.log > gpurun_out/$n.json; python - gpurun_out/$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], d["value"], d["unit"], d["ms_per_step"], "ms", d["dtype"], d["config"].get("launches_per_step"), (d.get("roofline", {}).get("whole_step") or {}).get("frac"))
PY
}
one final_bf16_w32 --workload hrnet_w32_256x192 --dtype bf16 --steps 20 --warmup 5
one final_bf16_w48 --workload hrnet_w48_384x288 --dtype bf16 --steps 20 --warmup 5
one final_w48 --workload hrnet_w48_384x288 --steps 20 --warmup 5 --no-cpu-baseline
one final_rsg --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline
one final_rsg48 --workload rsgnet_w48_384x288_coco --steps 10 --warmup 3 --no-cpu-baseline
one final_cp --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline
one final_train_hrnet --train --steps 10 --warmup 3
one final_train_rsgnet --train --workload rsgnet_w32_256x192_coco --steps 10 --warmup 3 --no-cpu-baseline
echo visit done
