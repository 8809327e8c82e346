#!/bin/bash
# stamp diagnostics of grouped fuse launches: OPS="name name ..."
mkdir -p gpurun_out
for op in ${OPS:-stage3.1.fuse.level0 stage4.0.fuse.level0}; do
  OP=$op timeout -k 10 400 python tools/conv_group_stamps.py > gpurun_out/stamps_$op.txt 2>&1 || { tail -20 gpurun_out/stamps_$op.txt; exit 1; }
  cat gpurun_out/stamps_$op.txt | cut -c1-260
done
