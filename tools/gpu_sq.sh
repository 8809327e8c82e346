#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
rm -rf gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
ls -la gpurun_out/pmc_sq | tail -3
