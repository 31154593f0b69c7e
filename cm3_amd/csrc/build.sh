#!/bin/bash
# Builds libcm3_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libcm3_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function ${CM3_EXTRA_FLAGS:-}"
OUT="${CM3_OUT:-${OUT}}"
mkdir -p "${HERE}/_obj"
# identity of the sources this library is built from (cm3_source_id(); cm3_amd/_lib.py refuses a library whose id differs from
# the sources next to it: a test run against a stale build proves nothing)
# (names in byte order, as _lib.source_id() sorts them; quoted throughout: a checkout path may contain blanks)
SRC_ID="$(cd "${HERE}" && { for f in $(LC_ALL=C ls *.hip *.h | LC_ALL=C sort); do cat "./${f}"; done; cat "../../include/cm3_amd.h"; } | sha256sum | cut -c1-16)"
pids=()
"${HIPCC}" ${FLAGS} -mllvm -amdgpu-kernarg-preload-count=16 -DCM3_PARTICLE_F32 -c "${HERE}/particle.hip" -o "${HERE}/_obj/particle_f32.o" &
pids+=($!)
"${HIPCC}" ${FLAGS} -DCM3_PARTICLE_F64 -c "${HERE}/particle.hip" -o "${HERE}/_obj/particle_f64.o" &
pids+=($!)
# the two shared-env float32 step kernels once more, scheduled for instruction-level parallelism (see the head of particle.hip)
"${HIPCC}" ${FLAGS} -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=${CM3_HOT_SCHED:-max-ilp} ${CM3_HOT_FLAGS:-} -DCM3_PARTICLE_F32 -DCM3_PARTICLE_ILP_TU \
  -c "${HERE}/particle.hip" -o "${HERE}/_obj/particle_f32_ilp.o" &
pids+=($!)
# Checkers: max-ILP scheduling throughout (C3 4.26 -> 4.16 us per tick; 2^16 .. 2^20 envs within 1 %)
"${HIPCC}" ${FLAGS} -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=${CM3_HOT_SCHED:-max-ilp} ${CM3_HOT_FLAGS:-} -c "${HERE}/checkers.hip" -o "${HERE}/_obj/checkers.o" &
pids+=($!)
for f in util advantage; do
  "${HIPCC}" ${FLAGS} -DCM3_SOURCE_ID="\"${SRC_ID}\"" -c "${HERE}/${f}.hip" -o "${HERE}/_obj/${f}.o" &
  pids+=($!)
done
# the matrix-core kernels: accumulators in the architectural VGPRs (gfx90a and later take them there), so that the layers' epilogues
# read them directly instead of through one v_accvgpr_read per value (60 per tick of the policy rollout), and no kernel is left
# above 256 registers (one wave per SIMD): k_ck_actor_x3 304 -> 236, k_policy_rollout<8, ., 4> 296 -> 251
for f in actor actor_checkers policy; do
  "${HIPCC}" ${FLAGS} -mllvm -amdgpu-mfma-vgpr-form=${CM3_MFMA_VGPR:-1} -DCM3_SOURCE_ID="\"${SRC_ID}\"" -c "${HERE}/${f}.hip" -o "${HERE}/_obj/${f}.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${HERE}/_obj/particle_f32.o" "${HERE}/_obj/particle_f32_ilp.o" "${HERE}/_obj/particle_f64.o" \
  "${HERE}/_obj/checkers.o" "${HERE}/_obj/util.o" "${HERE}/_obj/advantage.o" "${HERE}/_obj/actor.o" "${HERE}/_obj/actor_checkers.o" "${HERE}/_obj/policy.o"
echo "built ${OUT}"
