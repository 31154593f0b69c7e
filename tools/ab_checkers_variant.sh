#!/bin/bash
# Builds tools/variants/libcm3_hip_<name>.so = the product's objects with checkers.o recompiled under extra flags (same-box A/B of a
# Checkers kernel variant through CM3_AMD_LIB; run cm3_amd/csrc/build.sh first).   tools/ab_checkers_variant.sh <name> <flags...>
set -eu
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"; cd "$R"
NAME="$1"; shift
H="$R/cm3_amd/csrc"; O="$H/_obj"; T="$R/tools/_ab"; mkdir -p "$T"
SRC_ID="$(python3 -c 'import sys; sys.path.insert(0, "."); from cm3_amd._lib import source_id; print(source_id())')"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c "$H/checkers.hip" -o "$T/checkers_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/tools/variants/libcm3_hip_$NAME.so" "$O/particle_f32.o" "$O/particle_f32_ilp.o" "$O/particle_f64.o" \
  "$T/checkers_$NAME.o" "$O/util.o" "$O/advantage.o" "$O/batch.o" "$O/actor.o" "$O/actor_checkers.o" "$O/policy.o"
echo "built tools/variants/libcm3_hip_$NAME.so"
