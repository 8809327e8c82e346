#!/bin/bash
# round-5 closing visit, in two calls on one build (the whole of it no longer fits one 20-minute call):
#   PART=tests  every GPU test, then smoke()
#   PART=evidence  the final evidence set (TAG, default r05z)
# the library build of each part is in gpurun_out/<tag>_library_build.txt / build.log: they must agree
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -30 gpurun_out/build.log; exit 1; }
if [ "${PART:-tests}" = tests ]; then
  timeout -k 10 1050 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/pytest_gpu.log 2>&1
  rc=$?; grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2 | cut -c1-200; [ $rc -ne 0 ] && { grep -E "FAILED|Error|assert" gpurun_out/pytest_gpu.log | tail -20; exit $rc; }
  timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG:-r05z}_smoke.txt 2>&1 || { tail -20 gpurun_out/${TAG:-r05z}_smoke.txt; exit 1; }
  grep -E "smoke" gpurun_out/${TAG:-r05z}_smoke.txt | cut -c1-200
  tail -1 gpurun_out/build.log | cut -c1-60
else
  TAG=${TAG:-r05z} bash tools/gpu_r05_final.sh
fi
