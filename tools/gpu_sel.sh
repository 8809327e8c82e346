#!/bin/bash
# selected GPU tests in measuring mode: SEL="<pytest args>"
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
RSGNET_TEST_MEASURE=${MEASURE:-1} timeout -k 10 ${TEST_TIMEOUT:-800} python -m pytest ${SEL:-tests} ${K:+-k "$K"} -m gpu -q ${X--x} > gpurun_out/pytest_sel.log 2>&1
rc=$?; tail -${TAIL:-15} gpurun_out/pytest_sel.log; grep -n "rsgnet training step" gpurun_out/pytest_sel.log | cut -c1-400; exit $rc
