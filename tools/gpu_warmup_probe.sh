#!/bin/bash
# does the driver's (5 warm-up, 20 timed) region sit on the clock ramp?  same box: cold region (--settle-seconds 0), the
# default (settle_clocks), and long regions
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['dtype'], d['ms_per_step'], 'ms', d['value'], 'img/s; cold_start', d['config'].get('cold_start'))"; }
for rep in 1 2; do
  for dt in fp16 bf16; do
    python bench.py --dtype $dt --warmup 5 --steps 20 --no-cpu-baseline --settle-seconds 0 2>/dev/null | show "cold 5/20:"
    python bench.py --dtype $dt --warmup 5 --steps 20 --no-cpu-baseline 2>/dev/null | show "settled 5/20:"
    python bench.py --dtype $dt --warmup 200 --steps 1000 --no-cpu-baseline --settle-seconds 0 2>/dev/null | show "cold 200/1000:"
  done
done
