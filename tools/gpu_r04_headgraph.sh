#!/bin/bash
# GPU box: the RSGNet training head as HIP graphs -- probe (capture stages with backtraces), the bit-identity test, and the
# training step timed both ways at the bench's batch.
mkdir -p gpurun_out
timeout -k 10 600 python tools/head_capture_probe.py > gpurun_out/head_capture.log 2>&1 || { tail -40 gpurun_out/head_capture.log; exit 1; }
grep -v "Warn\|run_backward\|amdgpu.ids" gpurun_out/head_capture.log | cut -c1-200
timeout -k 10 600 python -m pytest tests/test_train_step.py -m gpu -x -q -k "head_replayed or training_call" > gpurun_out/head_graph_test.log 2>&1 || { tail -40 gpurun_out/head_graph_test.log; exit 1; }
tail -3 gpurun_out/head_graph_test.log
for mode in 0 1 0 1; do
  RSGNET_HIP_TRAIN="head_graph=$mode" timeout -k 10 400 python bench.py --train --workload rsgnet_w32_256x192_coco --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/head_graph_bench_$mode.log 2>&1 || { tail -20 gpurun_out/head_graph_bench_$mode.log; exit 1; }
  echo "head_graph=$mode: $(tail -1 gpurun_out/head_graph_bench_$mode.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d["config"])')"
done
