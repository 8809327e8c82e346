#!/bin/bash
# same-visit: 1..4 alternating streams on the headline config; chain pack forced on / off with two streams
mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'streams', d['config']['streams'], d['ms_per_step'], 'ms', d['value'], 'img/s whole', d['roofline']['whole_step']['frac'])"; }
for s in 1 2 3 4 2 3; do python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams $s 2>/dev/null | show hrnet_w32; done | tee gpurun_out/r04_streams.txt
for pk in 0 1 0 1; do RSGNET_HIP_PLAN="chain_pack=$pk" python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams 2 2>/dev/null | show "pack=$pk"; done | tee -a gpurun_out/r04_streams.txt
