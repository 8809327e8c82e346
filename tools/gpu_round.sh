#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (+ per-op table).  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 ${TEST_TIMEOUT:-500} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -40 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --per-op gpurun_out/per_op.json > gpurun_out/bench.log 2>&1
rc=$?
tail -5 gpurun_out/bench.log
exit $rc
