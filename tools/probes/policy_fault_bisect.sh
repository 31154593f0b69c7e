#!/bin/bash
# Round 5, second step: WHERE in the physics do the copies in lanes 48..63 part from the row's own lane?  Running hashes (two
# registers) over the values of each stage, compared across the four copies at the end of the section; counters by (stage, 16-lane row).
set -eu
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
SRC_COMMIT="${SRC_COMMIT:-c49681b}"
W="$R/tools/_pf"; mkdir -p "$W"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
OBJ="$R/cm3_amd/csrc/_obj"
fresh() {
  rm -rf "$1"; mkdir -p "$1/csrc" "$1/include"
  for f in $(git ls-tree --name-only "$SRC_COMMIT" cm3_amd/csrc/); do git show "$SRC_COMMIT:$f" > "$1/csrc/$(basename "$f")"; done
  git show "$SRC_COMMIT:include/cm3_amd.h" > "$1/include/cm3_amd.h"
  sed -i 's#"../../include/cm3_amd.h"#"../include/cm3_amd.h"#' "$1/csrc/common.h"
}
build() {
  local v="$1" form="$2"; shift 2
  "$HIPCC" $FLAGS -mllvm -amdgpu-mfma-vgpr-form="$form" -DCM3_SOURCE_ID="\"pf_$v\"" "$@" -Rpass-analysis=kernel-resource-usage \
      -c "$W/$v/csrc/policy.hip" -o "$W/$v/policy.o" 2> "$W/$v/resource_usage.txt"
  "$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$R/cm3_amd/libcm3_hip_pf_$v.so" "$OBJ/particle_f32.o" "$OBJ/particle_f32_ilp.o" \
      "$OBJ/particle_f64.o" "$OBJ/checkers.o" "$OBJ/util.o" "$OBJ/advantage.o" "$OBJ/batch.o" "$OBJ/actor.o" "$OBJ/actor_checkers.o" "$W/$v/policy.o"
  echo "== $v"; python3 tools/probes/resource_usage.py "$W/$v/resource_usage.txt" k_policy_rolloutILi8ELi2ELi4
}
for v in "$@"; do
  fresh "$W/$v"
  python3 tools/probes/policy_fault_patch.py "$v" "$W/$v/csrc"
  extra=""
  if [ "$v" = nopk ]; then extra="-Xclang -target-feature -Xclang -packed-fp32-ops"; fi
  if [ "$v" = noslp ]; then extra="-fno-slp-vectorize"; fi
  case "$v" in cut*A*) extra="-fno-slp-vectorize";; esac
  build "$v" 1 $extra &
done
wait
