#!/bin/bash
# training tests (normal bars) + the round's other-config numbers: training benches, bf16 lines, configs[2]/[4] with rocprof
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_train_step.py tests/test_train_tail.py -m gpu -q -x > gpurun_out/pytest_train.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/pytest_train.log
timeout -k 10 300 python bench.py --train --steps 10 --warmup 3 > gpurun_out/bench_train_hrnet.log 2>&1; tail -1 gpurun_out/bench_train_hrnet.log | cut -c1-300
timeout -k 10 300 python bench.py --train --workload rsgnet_w32_256x192_coco --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_train_rsgnet.log 2>&1; tail -1 gpurun_out/bench_train_rsgnet.log | cut -c1-300
for wl in hrnet_w32_256x192 hrnet_w48_384x288; do
  timeout -k 10 300 python bench.py --workload $wl --dtype bf16 --steps 20 --warmup 5 > gpurun_out/bench_bf16_$wl.log 2>&1; tail -1 gpurun_out/bench_bf16_$wl.log | cut -c1-260
done
timeout -k 10 300 python bench.py --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline --per-op gpurun_out/per_op_rsg.json > gpurun_out/bench_rsg.log 2>&1; tail -1 gpurun_out/bench_rsg.log | cut -c1-200
timeout -k 10 300 python bench.py --workload rsgnet_w48_384x288_coco --steps 10 --warmup 3 --no-cpu-baseline --per-op gpurun_out/per_op_rsg48.json > gpurun_out/bench_rsg48.log 2>&1; tail -1 gpurun_out/bench_rsg48.log | cut -c1-200
bash tools/gpu_cfg5.sh
echo visit done
