#!/bin/bash
# round 5: the persistent stem (conv2's weights resident in registers): parity through the model, then a same-visit A/B against stem_kernel
mkdir -p gpurun_out/r05e
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05e/build.log 2>&1 || { tail -30 gpurun_out/r05e/build.log; exit 1; }
timeout -k 10 800 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py tests/test_gpu_e2e.py -k "stem or golden or random_init or full_size or w48 or bf16" > gpurun_out/r05e/pytest.log 2>&1
rc=$?; tail -4 gpurun_out/r05e/pytest.log | cut -c1-250; [ $rc -ne 0 ] && exit $rc
bash tools/gpu_ab_env2.sh RSGNET_HIP_TUNE "stem_persist=0 stem_persist=1"
