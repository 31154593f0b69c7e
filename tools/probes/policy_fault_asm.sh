#!/bin/bash
# Round 5: edits of the failing build's ASSEMBLY (one instruction at a time) -- tools/probes/policy_fault_asm.py names the edits.
# device code of the old sources -> .s -> edit -> assemble -> link -> bundle -> host object with that device binary -> library.
set -eu
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
W="$R/tools/_pf"; LLVM=/opt/rocm/lib/llvm/bin; OBJ="$R/cm3_amd/csrc/_obj"
SRC="$W/ctrl/csrc"      # made by policy_fault_variants.sh / policy_fault_bisect.sh ctrl2
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -DCM3_SOURCE_ID=\"pf_asm\""
mkdir -p "$W/asm"
[ -f "$W/asm/policy.s" ] || /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o "$W/asm/policy.s" "$SRC/policy.hip"
for v in "$@"; do
  d="$W/asm/$v"; mkdir -p "$d"
  python3 tools/probes/policy_fault_asm.py "$v" "$W/asm/policy.s" "$d/policy.s"
  "$LLVM/clang" -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$d/policy.s" -o "$d/dev.o"
  "$LLVM/lld" -flavor gnu -m elf64_amdgpu --no-undefined -shared -o "$d/dev.out" "$d/dev.o"
  "$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 \
      -input=/dev/null -input="$d/dev.out" -output="$d/policy.hipfb"
  /opt/rocm/bin/hipcc $FLAGS --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$d/policy.hipfb" -c "$SRC/policy.hip" -o "$d/policy.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/cm3_amd/libcm3_hip_pf_asm_$v.so" "$OBJ/particle_f32.o" "$OBJ/particle_f32_ilp.o" \
      "$OBJ/particle_f64.o" "$OBJ/checkers.o" "$OBJ/util.o" "$OBJ/advantage.o" "$OBJ/batch.o" "$OBJ/actor.o" "$OBJ/actor_checkers.o" "$d/policy.o"
  echo "built asm_$v"
done
