#!/bin/bash
# round 5: the SGM graph-conv kernel (tests/test_train_ops.py) and every RSGNet test that now runs through it (plan build + training head)
mkdir -p gpurun_out/r05t
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05t/build.log 2>&1 || { tail -30 gpurun_out/r05t/build.log; exit 1; }
timeout -k 10 900 python -m pytest -m gpu -q tests/test_train_ops.py tests/test_gpu_e2e.py tests/test_train_step.py -k "sgm or rsgnet" > gpurun_out/r05t/pytest_sgm.log 2>&1
rc=$?; tail -6 gpurun_out/r05t/pytest_sgm.log | cut -c1-300; exit $rc
