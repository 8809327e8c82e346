#!/bin/bash
# Round 4: phase stamps of the grouped chain launch (as the plan launches it) + the headline bench of the same visit.
mkdir -p gpurun_out
python tools/chain_phase_stamps.py > gpurun_out/r04_chain_phase_stamps.txt 2>&1 || { tail -20 gpurun_out/r04_chain_phase_stamps.txt; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_base.json 2> gpurun_out/r04_bench_base.err
tail -c 1500 gpurun_out/r04_bench_base.json
