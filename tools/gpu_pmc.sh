#!/bin/bash
# kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE in their own runs) of the bench command
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --per-op gpurun_out/per_op.json > gpurun_out/bench.log 2>&1 || { tail -5 gpurun_out/bench.log; exit 1; }
tail -1 gpurun_out/bench.log | cut -c1-400
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof.log 2>&1 || { tail -5 gpurun_out/prof.log; exit 1; }
find gpurun_out/prof -name '*kernel_trace.csv' -delete
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1 || { tail -5 gpurun_out/pmc_write.log; exit 1; }
ls -la gpurun_out/pmc_fetch gpurun_out/pmc_write | head -12
