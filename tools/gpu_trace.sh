#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
rm -rf gpurun_out/prof4
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof4 -o trace -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/prof4.log 2>&1
tail -1 gpurun_out/prof4.log | cut -c1-160
python tools/trace_gaps.py gpurun_out/prof4 | tee gpurun_out/trace_gaps.txt
find gpurun_out/prof4 -name "*.csv" -size +2M -delete
