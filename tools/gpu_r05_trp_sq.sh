#!/bin/bash
# TRP tests, then the SQ counters of the CrowdPose flip test (bank conflicts of trp_kernel) and its kernel stats
mkdir -p gpurun_out/r05i; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05i/build.log 2>&1 || { tail -30 gpurun_out/r05i/build.log; exit 1; }
timeout -k 10 800 python -m pytest -m gpu -q -x tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_train_ops.py tests/test_train_step.py -k "trp or rsgnet" > gpurun_out/r05i/pytest.log 2>&1
rc=$?; tail -3 gpurun_out/r05i/pytest.log | cut -c1-200; [ $rc -ne 0 ] && exit $rc
rm -rf gpurun_out/pmc_sq gpurun_out/prof_cp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r05i/pmc_sq_cp.log 2>&1 || { tail -5 gpurun_out/r05i/pmc_sq_cp.log; exit 1; }
python tools/sq_summary.py r05i_tmp > /dev/null && mv profiles/r05i_tmp_sq_summary.json gpurun_out/r05i/sq_summary_cp.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05i/sq_summary_cp.json"))
for k, v in d.items():
    if "trp" in k or "bilinear" in k: print(k, v)
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cp -o trace -- python3 bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05i/prof_cp.log 2>&1 || { tail -5 gpurun_out/r05i/prof_cp.log; exit 1; }
grep -E "trp_|bilinear" $(find gpurun_out/prof_cp -name '*kernel_stats.csv' | head -1) | cut -c1-150
rm -rf gpurun_out/pmc_sq gpurun_out/prof_cp
