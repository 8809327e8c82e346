#!/bin/bash
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
for pad in 1; do for grp in 0 1; do
  RSGNET_HIP_TUNE="chain_pitch_pad=$pad" RSGNET_HIP_PLAN="group=$grp" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --per-op gpurun_out/po_${pad}_${grp}.json > gpurun_out/ab.log 2>&1
  python - <<PY
import json
d=json.load(open('gpurun_out/po_${pad}_${grp}.json'))
line=json.loads(open('gpurun_out/ab.log').read().strip().splitlines()[-1])
sel=[o for o in d['ops'] if o['op'].startswith('stage4.0') and ('chain' in o['op'])]
print('pad=${pad} group=${grp} ms/step', line['ms_per_step'], ' '.join(f"{o['op'].split('.',2)[2]}={o['us']}" for o in sel))
PY
done; done
