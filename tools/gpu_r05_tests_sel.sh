#!/bin/bash
# selected GPU tests: SEL="pytest selection args"
mkdir -p gpurun_out/r05t
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05t/build.log 2>&1 || { tail -30 gpurun_out/r05t/build.log; exit 1; }
timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest -m gpu -q -s ${SEL} > gpurun_out/r05t/pytest_sel.log 2>&1
rc=$?; grep -E "passed|failed|FAILED|Error|assert|sigma|err " gpurun_out/r05t/pytest_sel.log | cut -c1-300 | tail -${TAIL:-60}; exit $rc
