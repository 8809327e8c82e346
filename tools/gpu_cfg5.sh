#!/bin/bash
# BASELINE configs[2] and configs[4] on one GPU: HRNet-W48 384x288 bench; CrowdPose RSGNet flip-test bench + rocprof
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
python bench.py --workload hrnet_w48_384x288 --steps 10 --warmup 3 --no-cpu-baseline --per-op gpurun_out/per_op_w48.json > gpurun_out/bench_w48.log 2>&1; tail -1 gpurun_out/bench_w48.log | cut -c1-250
python bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline --per-op gpurun_out/per_op_cp.json > gpurun_out/bench_cp.log 2>&1; tail -1 gpurun_out/bench_cp.log | cut -c1-250
rm -rf gpurun_out/prof_cp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cp -o trace -- python bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_cp.log 2>&1
find gpurun_out/prof_cp -name '*kernel_trace.csv' -delete
head -12 gpurun_out/prof_cp/trace_kernel_stats.csv | cut -c1-150
