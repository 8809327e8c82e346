#!/bin/bash
# counter passes of the final build, ONE stream, cold region only (the counters serialise the kernels anyway):
# FETCH_SIZE, WRITE_SIZE each in a run of its own, then the SQ pass; summarise here afterwards with
#   python tools/pmc_summary.py r03p && python tools/sq_summary.py r03p
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
A="--steps 4 --warmup 2 --no-cpu-baseline --streams 1 --settle-seconds 0"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python bench.py $A > gpurun_out/pmc_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python bench.py $A > gpurun_out/pmc_write.log 2>&1 || { tail -5 gpurun_out/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python bench.py $A > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
ls -la gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq | grep csv
