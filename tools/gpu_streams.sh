#!/bin/bash
# same-visit: 1..4 alternating streams on the headline and the other inference configs; rocprofv3 stats of the driver's command
mkdir -p gpurun_out/r03n; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
O=gpurun_out/r03n
show() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 streams', d['config']['streams'], d['dtype'], d['ms_per_step'], 'ms', d['value'], 'img/s whole', d['roofline']['whole_step']['frac'], 'single_stream', (d.get('single_stream') or {}).get('ms_per_step'))"; }
for s in 1 2 1 2 3 4; do python bench.py --steps 200 --warmup 50 --no-cpu-baseline --streams $s 2>/dev/null | show hrnet_w32; done | tee $O/streams.txt
for w in hrnet_w48_384x288 rsgnet_w32_256x192_coco rsgnet_w48_384x288_coco; do
  for s in 1 2; do python bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --streams $s 2>/dev/null | show $w; done
done | tee -a $O/streams.txt
for s in 1 2; do python bench.py --workload rsgnet_w32_256x192_crowdpose --flip-test --steps 40 --warmup 10 --no-cpu-baseline --streams $s 2>/dev/null | show crowdpose_flip_test; done | tee -a $O/streams.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-240 $O/bench.json
rm -rf $O/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o trace -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
find $O/prof -name "*.csv" -size +1M -delete
head -12 $O/kernel_stats.csv | cut -c1-150
echo visit done
