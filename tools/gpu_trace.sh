#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
python - <<'PY' > gpurun_out/host_time.txt 2>&1
import sys, time, torch
sys.path.insert(0, '.')
import bench
cfg, model, sd, wh, batch = bench.make_model("hrnet_w32_256x192")
from rsgnet_amd.weights import synth_input
model = model.cuda(); x = synth_input(64, wh, 0).cuda()
with torch.no_grad():
    for _ in range(3): model(x)
    torch.cuda.synchronize()
    plan = model.plan_for(x)
    t0 = time.perf_counter()
    for _ in range(10): plan.launch_all()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host launch_all: {(t1-t0)/10*1e3:.3f} ms per step; with drain {(t2-t0)/10*1e3:.3f} ms; lanes {plan.n_lanes}, ops {len(plan.ops)}")
PY
cat gpurun_out/host_time.txt | tail -2
rm -rf gpurun_out/prof4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof4 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/prof4.log 2>&1
tail -2 gpurun_out/prof4.log | cut -c1-200
ls -la gpurun_out/prof4
