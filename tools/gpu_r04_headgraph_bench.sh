for b in 8 32; do for mode in 0 1; do
  RSGNET_HIP_TRAIN="head_graph=$mode" timeout -k 10 400 python bench.py --train --workload rsgnet_w32_256x192_coco --batch $b --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/head_graph_bench_b${b}_$mode.log 2>&1 || { tail -20 gpurun_out/head_graph_bench_b${b}_$mode.log; exit 1; }
  echo "batch $b head_graph=$mode: $(tail -1 gpurun_out/head_graph_bench_b${b}_$mode.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["unit"], d["ms_per_step"], "ms/step", d["config"]["loss_first_last"])')"
done; done
