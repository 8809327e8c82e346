#!/bin/bash
# The driver's round-end check: every GPU test, normal bars.
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -8 gpurun_out/pytest_gpu.log; exit $rc
