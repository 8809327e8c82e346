#!/bin/bash
# same-visit sweep of the weight-gradient policy knobs (RSGNET_HIP_WGRAD_MIN_PX / _QUAD_MIN_WGS / _LDS / _PX_CAP) on the training step
mkdir -p gpurun_out
one() { python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
CFGS=${CFGS:-0,0,73728 0,0,49152 0,0,36864 0,0,24576 0,0,98304 0,0,73728}
for cfg in $CFGS; do
  IFS=, read a b c d e <<< "$cfg"
  echo -n "min_px=$a quad_min_wgs=$b lds=$c px_cap=${d:-0} waves=${e:-0}: "; RSGNET_HIP_WGRAD_MIN_PX=$a RSGNET_HIP_WGRAD_QUAD_MIN_WGS=$b RSGNET_HIP_WGRAD_LDS=$c RSGNET_HIP_WGRAD_PX_CAP=${d:-0} RSGNET_HIP_WGRAD_WAVES=${e:-0} one
done
