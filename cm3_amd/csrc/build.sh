#!/bin/bash
# Builds libcm3_hip.so for gfx950 in-tree (cross-compiles without a GPU).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libcm3_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function ${CM3_EXTRA_FLAGS:-}"
OUT="${CM3_OUT:-${OUT}}"
OBJ="${CM3_OBJ_DIR:-${HERE}/_obj}"   # (variant builds -- span stamps, A/B flags -- keep their objects out of the product's)
mkdir -p "${OBJ}"
# No SLP vectorisation in the translation units that hold the float32 physics (round 5, profiles/r05_policy_fault.txt): the
# vectoriser packs pairs of unrelated scalar multiplies into v_pk_mul_f32 and then broadcasts ONE element of such a pair with a
# cross-half select (`op_sel:[0,1]`: the low result takes the high dword of a source).  On MI355X exactly that instruction returned
# +-0 as its low result in lanes 48..63, a handful of times per launch, while the SIMD's other wave ran a float16 matrix layer -- no
# wait state around it helps, the same product without a cross-half select is clean.  Cost: 1379 -> 1384 instructions in the C2 step
# kernel.  tools/isa_lint.py (run below) fails the build if a packed float32 instruction with a low-half cross select is left.
PHYS="${CM3_PHYS_FLAGS:--fno-slp-vectorize}"
# identity of the sources this library is built from (cm3_source_id(); cm3_amd/_lib.py refuses a library whose id differs from
# the sources next to it: a test run against a stale build proves nothing)
# (names in byte order, as _lib.source_id() sorts them; quoted throughout: a checkout path may contain blanks)
SRC_ID="$(cd "${HERE}" && { for f in $(LC_ALL=C ls *.hip *.h | LC_ALL=C sort); do cat "./${f}"; done; cat "../../include/cm3_amd.h"; } | sha256sum | cut -c1-16)"
pids=()
"${HIPCC}" ${FLAGS} ${PHYS} -mllvm -amdgpu-kernarg-preload-count=16 -DCM3_PARTICLE_F32 -c "${HERE}/particle.hip" -o "${OBJ}/particle_f32.o" &
pids+=($!)
"${HIPCC}" ${FLAGS} ${PHYS} -DCM3_PARTICLE_F64 -c "${HERE}/particle.hip" -o "${OBJ}/particle_f64.o" &
pids+=($!)
# the two shared-env float32 step kernels once more, scheduled for instruction-level parallelism (see the head of particle.hip)
"${HIPCC}" ${FLAGS} ${PHYS} -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=${CM3_HOT_SCHED:-max-ilp} ${CM3_HOT_FLAGS:-} -DCM3_PARTICLE_F32 -DCM3_PARTICLE_ILP_TU \
  -c "${HERE}/particle.hip" -o "${OBJ}/particle_f32_ilp.o" &
pids+=($!)
# Checkers: max-ILP scheduling throughout (C3 4.26 -> 4.16 us per tick; 2^16 .. 2^20 envs within 1 %)
"${HIPCC}" ${FLAGS} -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=${CM3_HOT_SCHED:-max-ilp} ${CM3_HOT_FLAGS:-} -c "${HERE}/checkers.hip" -o "${OBJ}/checkers.o" &
pids+=($!)
for f in util advantage batch; do
  "${HIPCC}" ${FLAGS} -DCM3_SOURCE_ID="\"${SRC_ID}\"" -c "${HERE}/${f}.hip" -o "${OBJ}/${f}.o" &
  pids+=($!)
done
# the matrix-core kernels (CM3_MATRIX_KERNEL in actor_common.h: two waves per SIMD declared, so the compiler's default selection
# keeps the accumulators in architectural VGPRs -- k_ck_actor_x3 236 registers, k_policy_rollout<8, ., 4> <= 256; round 4's
# experimental -amdgpu-mfma-vgpr-form=1 is gone.  CM3_MFMA_VGPR=0|1 still forces a form for A/B builds)
MFMA_FORM=""
if [ -n "${CM3_MFMA_VGPR:-}" ]; then MFMA_FORM="-mllvm -amdgpu-mfma-vgpr-form=${CM3_MFMA_VGPR}"; fi
for f in actor actor_checkers policy policy_checkers; do
  "${HIPCC}" ${FLAGS} ${PHYS} ${MFMA_FORM} -DCM3_SOURCE_ID="\"${SRC_ID}\"" -Rpass-analysis=kernel-resource-usage -c "${HERE}/${f}.hip" -o "${OBJ}/${f}.o" 2> "${OBJ}/${f}.resource_usage.txt" \
    || { grep -v "remark:" "${OBJ}/${f}.resource_usage.txt" >&2; exit 1; } &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
# the matrix units' diagnostics went to their resource-usage files with the remarks: show what is not a remark
for f in actor actor_checkers policy policy_checkers; do grep -v "remark:" "${OBJ}/${f}.resource_usage.txt" | grep -E "warning|error" -A3 >&2 || true; done
OBJS=("${OBJ}/particle_f32.o" "${OBJ}/particle_f32_ilp.o" "${OBJ}/particle_f64.o" "${OBJ}/checkers.o" "${OBJ}/util.o" "${OBJ}/advantage.o"
      "${OBJ}/batch.o" "${OBJ}/actor.o" "${OBJ}/actor_checkers.o" "${OBJ}/policy.o" "${OBJ}/policy_checkers.o")
if [ "${CM3_SKIP_ISA_LINT:-0}" != 1 ]; then
  python3 "${HERE}/../../tools/isa_lint.py" "${OBJS[@]}"     # (the objects of THIS build, not whatever else sits in ${OBJ})
fi
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${OBJS[@]}"
echo "built ${OUT}"
