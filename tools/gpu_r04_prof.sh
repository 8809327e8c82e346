#!/bin/bash
# Round 4 evidence for the headline config: the driver's bench line, rocprofv3 kernel stats of the ONE-stream command (what
# the per-kernel figures are quoted on), FETCH_SIZE / WRITE_SIZE in their own --pmc passes, one SQ pass.  TAG names the files.
TAG=${TAG:-r04}
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
python -c "from rsgnet_amd import build; print(build.describe())" > gpurun_out/${TAG}_library_build.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --per-op gpurun_out/${TAG}_per_op.json > gpurun_out/${TAG}_bench.log 2>&1 || { tail -5 gpurun_out/${TAG}_bench.log; exit 1; }
tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o trace -- python3 bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1 || { tail -5 gpurun_out/prof.log; exit 1; }
find gpurun_out/prof -name '*kernel_trace.csv' -delete
cp $(find gpurun_out/prof -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats_one_stream.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o pmc -- python3 bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/pmc_fetch.log 2>&1 || { tail -5 gpurun_out/pmc_fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o pmc -- python3 bench.py --steps 4 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/pmc_write.log 2>&1 || { tail -5 gpurun_out/pmc_write.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -o pmc -- python3 bench.py --steps 3 --warmup 2 --streams 1 --no-cpu-baseline > gpurun_out/pmc_sq.log 2>&1 || { tail -5 gpurun_out/pmc_sq.log; exit 1; }
mkdir -p profiles
python tools/pmc_summary.py ${TAG}_tmp > /dev/null && mv profiles/${TAG}_tmp_pmc_summary.json gpurun_out/${TAG}_pmc_summary.json
python tools/sq_summary.py ${TAG}_tmp > /dev/null && mv profiles/${TAG}_tmp_sq_summary.json gpurun_out/${TAG}_sq_summary.json
head -c 400 gpurun_out/${TAG}_bench.json; echo; head -12 gpurun_out/${TAG}_kernel_stats_one_stream.csv | cut -c1-160
# the raw counter / trace dumps stay on the box (gpurun copies back at most 64 MiB): the summaries above are what is kept
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq gpurun_out/prof gpurun_out/prof_cp gpurun_out/prof_w48
