#!/bin/bash
# round-5 closing visit: every GPU test, smoke(), then the final evidence set (TAG, default r05z) -- all on one build
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?; tail -2 gpurun_out/pytest_gpu.log | cut -c1-200; [ $rc -ne 0 ] && { grep -E "FAILED|Error|assert" gpurun_out/pytest_gpu.log | tail -20; exit $rc; }
timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG:-r05z}_smoke.txt 2>&1 || { tail -20 gpurun_out/${TAG:-r05z}_smoke.txt; exit 1; }
grep -E "smoke" gpurun_out/${TAG:-r05z}_smoke.txt | cut -c1-200
TAG=${TAG:-r05z} bash tools/gpu_r05_final.sh
