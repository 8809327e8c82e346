#!/bin/bash
# RSGNet training bench + kernel stats of the same command
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
python bench.py --train --workload rsgnet_w32_256x192_coco --steps 10 --warmup 3 > gpurun_out/bench_train_rsgnet.log 2>&1 || { tail -20 gpurun_out/bench_train_rsgnet.log; exit 1; }
tail -1 gpurun_out/bench_train_rsgnet.log | cut -c1-600
rm -rf gpurun_out/prof_rsgtrain
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rsgtrain -o rsgtrain -- python bench.py --train --workload rsgnet_w32_256x192_coco --steps 5 --warmup 2 > gpurun_out/prof_rsgtrain.log 2>&1 || { tail -20 gpurun_out/prof_rsgtrain.log; exit 1; }
f=$(find gpurun_out/prof_rsgtrain -name "*kernel_stats.csv" | head -1)
head -40 "$f" | cut -c1-220 > gpurun_out/rsgtrain_kernel_stats.txt
find gpurun_out/prof_rsgtrain -name "*.csv" ! -name "*kernel_stats.csv" -delete
find gpurun_out/prof_rsgtrain -name "*.db" -delete
head -30 gpurun_out/rsgtrain_kernel_stats.txt
