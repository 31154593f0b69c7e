#!/bin/bash
# What each part of a tick costs the whole-episode PARTICLE policy kernel at C2 (4096 envs x 4 agents, f16x3): variant libraries that differ from
# the product in ONE probe macro of policy.hip / actor.hip (results wrong by construction; only the time matters).
#   bash tools/r6/policy_whatif.sh build ; gpurun -- 'bash tools/r6/policy_whatif.sh run'
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
V="${WHATIF:-nophasea:-DCM3_PROBE_P_NO_PHASEA nophaseb:-DCM3_PROBE_P_NO_PHASEB nohead:-DCM3_PROBE_P_NO_HEAD nophys:-DCM3_PROBE_P_NO_PHYS nostores:-DCM3_PROBE_P_NO_STORES norows:-DCM3_PROBE_P_NO_ROWS noreset:-DCM3_PROBE_P_NO_RESET ks1:-DCM3_PROBE_P_PHASEA_KS1}"
if [ "${1:-run}" = build ]; then
  mkdir -p tools/variants /tmp/obj_pwhatif
  O="$R/cm3_amd/csrc/_obj"
  SRC_ID="$(cd cm3_amd/csrc && { for f in $(LC_ALL=C ls *.hip *.h | LC_ALL=C sort); do cat "./${f}"; done; cat "../../include/cm3_amd.h"; } | sha256sum | cut -c1-16)"
  for v in $V; do
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -fno-slp-vectorize $f -DCM3_SOURCE_ID="\"${SRC_ID}\"" \
      -c cm3_amd/csrc/policy.hip -o /tmp/obj_pwhatif/pol_$n.o &
  done; wait
  for v in $V; do
    n=${v%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libcm3_hip_$n.so "$O/particle_f32.o" "$O/particle_f32_ilp.o" "$O/particle_f64.o" \
      "$O/checkers.o" "$O/util.o" "$O/advantage.o" "$O/batch.o" "$O/actor.o" "$O/actor_checkers.o" "$O/policy_checkers.o" /tmp/obj_pwhatif/pol_$n.o && echo "built $n"
  done
else
  cd "${GRAFT_REPO_ROOT:-$R}"
  for rep in 1 2; do
    for v in product $V; do
      n=${v%%:*}; lib=""; [ $n != product ] && lib="$PWD/tools/variants/libcm3_hip_$n.so"
      echo "$n $(CM3_AMD_LIB=$lib CM3_AMD_ALLOW_STALE=1 python tools/policy_row_tiles.py --worker ${CASE:-particle_stage2_antipodal 4 4096} f16x3 2>/dev/null | tail -1)"
    done
  done
fi
