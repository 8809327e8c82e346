#!/bin/bash
# run a pytest selection on the GPU box: ARGS="tests/test_x.py -k name"
mkdir -p gpurun_out/pt
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/pt/build.log 2>&1 || { tail -30 gpurun_out/pt/build.log; exit 1; }
eval "timeout -k 10 ${TEST_TIMEOUT:-900} python -m pytest -m gpu -q $ARGS" > gpurun_out/pt/pytest.log 2>&1
rc=$?; tail -${TAIL:-12} gpurun_out/pt/pytest.log | cut -c1-300; exit $rc
