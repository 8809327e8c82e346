#!/bin/bash
mkdir -p gpurun_out/tt; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
rm -rf gpurun_out/tt/prof
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tt/prof -o trace -- python bench.py --train --steps 8 --warmup 4 --no-cpu-baseline ${TRAIN_ARGS} > gpurun_out/tt/prof.log 2>&1 || { tail -5 gpurun_out/tt/prof.log; exit 1; }
tail -1 gpurun_out/tt/prof.log | cut -c1-200
python tools/trace_train_gaps.py gpurun_out/tt/prof | tee gpurun_out/tt/gaps.txt
find gpurun_out/tt/prof -name "*.csv" -size +2M -delete
