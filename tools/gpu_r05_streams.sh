#!/bin/bash
# round 5, same visit: 1..4 alternating streams on the headline (the band launch of the fuse layers holds a CU's LDS: does two stay best?)
mkdir -p gpurun_out/r05s; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r05s/build.log 2>&1 || exit 1
O=gpurun_out/r05s
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 streams', d['config']['streams'], d['dtype'], d['ms_per_step'], 'ms', d['value'], 'img/s whole', d['roofline']['whole_step']['frac'], 'single_stream', (d.get('single_stream') or {}).get('ms_per_step'))"; }
for s in 1 2 3 4 1 2 3 4; do python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams $s 2>/dev/null | show hrnet_w32; done | tee $O/streams.txt
