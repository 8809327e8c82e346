#!/bin/bash
# measurement pass of the parity figures the per-tensor bars are derived from (RSGNET_TEST_MEASURE=1: bars off)
mkdir -p gpurun_out; rm -f gpurun_out/parity.txt
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
RSGNET_TEST_MEASURE=1 timeout -k 10 900 python -m pytest tests/test_train_step.py tests/test_gpu_e2e.py -q -m gpu -k "rsgnet_training_step_matches or bf16_path or teacher_forced or rsgnet_w32" > gpurun_out/measure.log 2>&1
tail -5 gpurun_out/measure.log
