#!/bin/bash
# Per-kernel device time of the HRNet-W32 training step (BASELINE configs[3]) at batch 32 and batch 8: which kernels do not
# scale with the batch (the step is 15.4 ms at 32 crops and 11.3 ms at 8 for RSGNet).
mkdir -p gpurun_out; export TMPDIR=/tmp
for b in 32 8; do
  rm -rf gpurun_out/prof_train_$b
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_$b -o trace -- python3 bench.py --train --batch $b --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_train_$b.log 2>&1 || { tail -5 gpurun_out/prof_train_$b.log; exit 1; }
  find gpurun_out/prof_train_$b -name '*kernel_trace.csv' -delete
  cp $(find gpurun_out/prof_train_$b -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG:-r04}_train_hrnet_b${b}_kernel_stats.csv
  tail -1 gpurun_out/prof_train_$b.log | cut -c1-200
done
