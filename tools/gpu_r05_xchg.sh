#!/bin/bash
# halo exchange inside the launch: its tests, then the same-visit A/B (RSGNET_HIP_PLAN chain_exchange=0|auto)
set -o pipefail
ARGS='tests/test_gpu_kernels.py tests/test_gpu_e2e.py -k "chain or stream or handoff"' TAIL=6 bash tools/gpu_pytest.sh || exit 1
bash tools/gpu_ab_env2.sh RSGNET_HIP_PLAN "chain_exchange=0 chain_exchange=auto"
