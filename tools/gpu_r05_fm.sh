#!/bin/bash
# round 5: the band launch of the fuse layers (rsg_fuse_module) -- parity first, then a same-visit A/B against the grouped launches
set -o pipefail
OUT=gpurun_out/r05a
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -30 $OUT/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_fuse_module.py -x -q -s > $OUT/pytest_fm.log 2>&1
rc=$?; tail -25 $OUT/pytest_fm.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do
  for v in 0 1; do
    for st in 2 1; do
      env RSGNET_HIP_PLAN="fuse_module=$v" timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams $st \
        ${PEROP:+--per-op $OUT/per_op_${v}_${st}.json} > $OUT/fm_${v}_s${st}_${rep}.json 2> $OUT/fm_${v}_s${st}_${rep}.err || { tail -5 $OUT/fm_${v}_s${st}_${rep}.err; exit 1; }
      python - $OUT/fm_${v}_s${st}_${rep}.json "FUSE_MODULE=$v streams=$st run $rep" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
f = d["roofline"]["families"]
print(f"{sys.argv[2]}: {d['ms_per_step']:.4f} ms/step {d['value']:.0f} img/s launches {d['config']['launches_per_step']} | " +
      " | ".join(f"{k.split(' ')[0]} {x['us']:.0f}us" for k, x in f.items()), flush=True)
PY
    done
  done
done
