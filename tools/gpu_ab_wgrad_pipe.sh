one() { python bench.py --train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'; }
for cfg in "1 0" "1 -192" "1 -224" "1 -160" "0 0" "1 0" "1 -192"; do set -- $cfg; echo -n "pipe=$1 wgs=$2: "; RSGNET_HIP_TRAIN="wgrad_pipe=$1" RSGNET_HIP_TUNE="wgrad_min_px=$2" one; done
