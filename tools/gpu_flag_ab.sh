#!/bin/bash
# A/B of a build variant within one visit (same GPU): baseline build, then rebuild with RSGNET_HIP_NO_VGPR_FORM
mkdir -p gpurun_out
run() { for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' '; done; echo; }
python -c "from rsgnet_amd import build as b; b.build(force=True)" > gpurun_out/build.log 2>&1 || exit 1
echo "== baseline"; run
RSGNET_HIP_NO_VGPR_FORM="$1" python -c "from rsgnet_amd import build as b; b.build(force=True)" > gpurun_out/build2.log 2>&1 || { tail gpurun_out/build2.log; exit 1; }
echo "== vgpr-form on $1"; run
python -m pytest tests -m gpu -q -x -k "golden or chain or conv_igemm" 2>&1 | tail -2
