#!/bin/bash
# batches in flight: bench.py --streams 1 | 2 | 3 | 4, same visit (each stream has its own plan; side-stream plans are throughput plans)
mkdir -p gpurun_out/st
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/st/build.log 2>&1 || exit 1
for rep in 1 2; do for n in 2 3 4; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --streams $n > gpurun_out/st/s${n}_${rep}.json 2> gpurun_out/st/s${n}_${rep}.err || { tail -5 gpurun_out/st/s${n}_${rep}.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/st/s${n}_${rep}.json').read().strip().splitlines()[-1]); print('streams', $n, 'run', $rep, d['ms_per_step'], 'ms', d['value'], 'img/s', d.get('overlapped_outputs_bit_identical'))"
done; done
