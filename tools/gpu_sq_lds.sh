#!/bin/bash
# one --pmc pass of LDS-side SQ counters (what paces the chain kernel's k-loop?)
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
rm -rf gpurun_out/pmc_lds
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/pmc_lds -o pmc -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_lds.log 2>&1 || { tail -8 gpurun_out/pmc_lds.log; exit 1; }
ls gpurun_out/pmc_lds | tail -3
