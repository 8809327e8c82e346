#!/bin/bash
# round 5: the write-after-read build of DESIGN 9b' (tools/war_separate.py) beside the model graph, then the evidence pass of the current build
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { tail -20 gpurun_out/build.log; exit 1; }
TRP_OLD_VARIANTS=plain,war_separated,plain timeout -k 10 500 python tools/trp_old_projection_nop.py > gpurun_out/r05_trp_war_discriminator.txt 2>&1
rc=$?; cat gpurun_out/r05_trp_war_discriminator.txt | tail -8; [ $rc -ne 0 ] && exit $rc
TAG=${TAG:-r05a} bash tools/gpu_r04_prof.sh
