#!/bin/bash
# round 5 final evidence on the shipped build: headline (bench line, per-op, rocprofv3 one-stream stats, FETCH / WRITE / SQ passes),
# then the other BASELINE configs.  TAG names the files (default r05z).
export TAG=${TAG:-r05z}
bash tools/gpu_r04_prof.sh || exit 1
bash tools/gpu_r04_other.sh || exit 1
