#!/bin/bash
# bench + rocprofv3 kernel trace of the same command; summaries land in gpurun_out/ (copy to profiles/).
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1 || { tail -20 gpurun_out/bench.log; exit 1; }
tail -2 gpurun_out/bench.log
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof.log 2>&1
rc=$?
tail -3 gpurun_out/prof.log
find gpurun_out/prof -name '*kernel_stats*' | head -3
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -30 "$f"
# the raw per-dispatch trace can be large: keep only the stats
find gpurun_out/prof -name '*kernel_trace.csv' -size +20M -delete
exit $rc
