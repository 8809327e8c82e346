#!/bin/bash
# round 5: conv_dual on two waves per workgroup (doubled weight-fragment reuse): parity, then per-op and whole-step A/B against four waves
mkdir -p gpurun_out/r05g
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05g/build.log 2>&1 || { tail -30 gpurun_out/r05g/build.log; exit 1; }
timeout -k 10 600 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py tests/test_gpu_e2e.py -k "dual or golden" > gpurun_out/r05g/pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r05g/pytest.log | cut -c1-250; [ $rc -ne 0 ] && exit $rc
VALS="dual_waves=4 dual_waves=2" OPS=7 bash tools/gpu_r05_perop_tune.sh | cut -c1-400
bash tools/gpu_ab_env2.sh RSGNET_HIP_TUNE "dual_waves=4 dual_waves=2"
