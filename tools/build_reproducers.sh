#!/bin/bash
# Rebuilds every binary under tools/_build/ from the sources in tools/ (they are build products: none is committed).
# The reproducers of DESIGN 9b' (the TRP projection fault beside another stream's kernels) and the LDS canaries.
#   tools/build_reproducers.sh            all of them
# Needs hipcc and ROCm's llvm tools (clang, ld.lld); no GPU.
set -euo pipefail
cd "$(dirname "$0")"
mkdir -p _build
HIPCC=${HIPCC:-hipcc}
LLVM=${LLVM:-/opt/rocm/lib/llvm/bin}
F="--offload-arch=gfx950 -std=c++17 -ffp-contract=fast -shared -fPIC"
$HIPCC $F -O3 -o _build/libtrp_old_plain.so trp_old_projection.hip
$HIPCC $F -O3 -mllvm -amdgpu-waitcnt-forcezero -o _build/libtrp_old_forcezero.so trp_old_projection.hip
$HIPCC $F -O1 -o _build/libtrp_old_o1.so trp_old_projection.hip
$HIPCC $F -O3 -DTWO_COPIES -o _build/libtrp_old_twocopies.so trp_old_projection.hip
$HIPCC $F -O2 -o _build/liblds_canary.so lds_canary.hip
$HIPCC $F -O2 -o _build/liblds_broadcast_canary.so lds_broadcast_canary.hip
# the code objects of tools/trp_old_projection_nop.py: the -O3 device assembly as it is, and with `s_nop 3` in front of EVERY
# v_pk_fma_f32 (four wait states between the v_mov_b32 that assembles an operand and the packed FMA that reads it)
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only -o _build/trp_old.s trp_old_projection.hip
sed 's/^\(\s*\)v_pk_fma_f32 /\1s_nop 3\n\1v_pk_fma_f32 /' _build/trp_old.s > _build/trp_old_nop.s
# round 5 (DESIGN 9b', the write-after-read pair): 16 wait states in front of every ds_read whose return overwrites a source
# register of a v_pk_fma_f32 issued at most three instructions earlier (tools/war_separate.py)
python3 war_separate.py _build/trp_old.s _build/trp_old_war.s
for v in "trp_old:trp_old_plain" "trp_old_nop:trp_old_nop_before_pk_fma" "trp_old_war:trp_old_war_separated"; do
  src=${v%%:*}; dst=${v##*:}
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c _build/$src.s -o _build/$dst.o
  $LLVM/ld.lld -shared _build/$dst.o -o _build/$dst.co
done
echo "plain: $(grep -c v_pk_fma_f32 _build/trp_old.s) v_pk_fma_f32; nop build: $(grep -c 's_nop 3' _build/trp_old_nop.s) s_nop 3 inserted"
ls -la _build/*.so _build/*.co
