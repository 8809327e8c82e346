#!/bin/bash
# Soak (round 4: the new chain bodies; round 5, TAG=r05: band launches, persistent stem, two-wave conv_dual, piece-major TRP staging, the halo exchange inside the chain launches of side-stream plans): one-stream replays bit-identical to the first, and forwards overlapping on two streams
# bit-identical to the one-stream result (the regime the headline value is measured in).
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || exit 1
O=gpurun_out/${TAG:-r04}_soak.txt; : > $O
timeout -k 10 500 python tools/soak_determinism.py 300 >> $O 2>&1 || { tail -5 $O; exit 1; }
for spec in "hrnet_w32_256x192 600 32" "hrnet_w32_256x192 600 1" "rsgnet_w32_256x192_coco 360 16" "rsgnet_w32_256x192_crowdpose 240 16" "hrnet_w48_384x288 120 8"; do
  set -- $spec
  timeout -k 10 400 python tools/stream_soak.py $1 $2 $3 >> $O 2>&1 || { tail -5 $O; exit 1; }
done
SOAK_DTYPE=bf16 timeout -k 10 400 python tools/stream_soak.py hrnet_w32_256x192 360 32 >> $O 2>&1 || { tail -5 $O; exit 1; }
grep -v "amdgpu.ids" $O
