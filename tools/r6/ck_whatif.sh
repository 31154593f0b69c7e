#!/bin/bash
# What each part of a tick costs the UNSTAMPED one-launch Checkers rollout: variant libraries that differ from the product in ONE
# probe macro of policy_checkers.hip (results are wrong by construction; only the time matters).  Build here, run on the GPU box:
#   bash tools/r6/ck_whatif.sh build ; gpurun -- 'bash tools/r6/ck_whatif.sh run'
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
V="${WHATIF:-skipenv:-DCM3_PROBE_SKIP_ENV h2ks1:-DCM3_PROBE_H2_KS=1 skipsmall:-DCM3_PROBE_SKIP_SMALL notable:-DCM3_PROBE_NO_TABLE skipconv:-DCM3_PROBE_SKIP_CONV skiplin:-DCM3_PROBE_SKIP_LIN skipself:-DCM3_PROBE_SKIP_SELF notail:-DCM3_PROBE_NO_TAIL_BARRIERS}"
if [ "${1:-run}" = build ]; then
  mkdir -p tools/variants /tmp/obj_whatif
  O="$R/cm3_amd/csrc/_obj"
  SRC_ID="$(cd cm3_amd/csrc && { for f in $(LC_ALL=C ls *.hip *.h | LC_ALL=C sort); do cat "./${f}"; done; cat "../../include/cm3_amd.h"; } | sha256sum | cut -c1-16)"
  for v in $V; do
    n=${v%%:*}; f=${v#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -fno-slp-vectorize $f -DCM3_SOURCE_ID="\"${SRC_ID}\"" \
      -c cm3_amd/csrc/policy_checkers.hip -o /tmp/obj_whatif/pc_$n.o &
  done; wait
  for v in $V; do
    n=${v%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libcm3_hip_$n.so "$O/particle_f32.o" "$O/particle_f32_ilp.o" "$O/particle_f64.o" \
      "$O/checkers.o" "$O/util.o" "$O/advantage.o" "$O/batch.o" "$O/actor.o" "$O/actor_checkers.o" "$O/policy.o" /tmp/obj_whatif/pc_$n.o && echo "built $n"
  done
else
  cd "${GRAFT_REPO_ROOT:-$R}"
  for rep in 1 2 3; do
    for v in product $V; do
      n=${v%%:*}; lib=""; [ $n != product ] && lib="$PWD/tools/variants/libcm3_hip_$n.so"
      echo "$n $(CM3_AMD_LIB=$lib CM3_AMD_ALLOW_STALE=1 python tools/ck_policy_worker.py 20 2>/dev/null | tail -1 | cut -c58-)"
    done
  done
fi
