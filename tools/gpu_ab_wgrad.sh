#!/bin/bash
# same-visit sweep of the weight-gradient policy knobs (RSGNET_HIP_TUNE wgrad_min_px / wgrad_quad_min_wgs / wgrad_lds / wgrad_px_cap) on the training step
mkdir -p gpurun_out
one() { python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
CFGS=${CFGS:-0,0,73728 0,0,49152 0,0,36864 0,0,24576 0,0,98304 0,0,73728}
for cfg in $CFGS; do
  IFS=, read a b c d e <<< "$cfg"
  echo -n "min_px=$a quad_min_wgs=$b lds=$c px_cap=${d:-0} waves=${e:-0}: "; RSGNET_HIP_TUNE="wgrad_min_px=$a,wgrad_quad_min_wgs=$b,wgrad_lds=$c,wgrad_px_cap=${d:-0},wgrad_waves=${e:-0}" one
done
