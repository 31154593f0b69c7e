#!/bin/bash
# Same-box comparison of two BUILDS of libcm3_hip.so (the method that caught this round's hidden regressions: a within-build
# A/B can show a "gain" against a path the change itself slowed down).
#
#   tools/ab_builds.sh <base-commit> "c2 trajectory" "c5 in-place" ...      (run on the GPU box, e.g. through gpurun)
#
# Builds tools/variants/libcm3_hip_base.so from <base-commit>'s cm3_amd/csrc + include (same ABI required) if it is not there yet
# (hipcc cross-compiles, so do that step in the build container), then alternates `bench.py --no-extras` between the two
# libraries (CM3_AMD_LIB selects the base one), three rounds, printing us per tick.  Remove the base library afterwards: it is a
# measurement artefact and must not travel with the product.
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"; cd "$R"
BASE="$1"; shift
LIB="$R/tools/variants/libcm3_hip_base.so"
if [ ! -f "$LIB" ]; then
  T=$(mktemp -d); mkdir -p "$T/csrc" "$T/include"
  for f in $(git ls-tree --name-only "$BASE" cm3_amd/csrc/); do git show "$BASE:$f" > "$T/csrc/$(basename "$f")"; done
  git show "$BASE:include/cm3_amd.h" > "$T/include/cm3_amd.h"
  sed -i 's#"../../include/cm3_amd.h"#"../include/cm3_amd.h"#' "$T/csrc/common.h"
  sed -i 's#OUT="${HERE}/../libcm3_hip.so"#OUT="${HERE}/../libcm3_hip_base.so"#' "$T/csrc/build.sh"
  bash "$T/csrc/build.sh" && cp "$T/libcm3_hip_base.so" "$LIB" || exit 1
fi
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
echo "workload mode build us/tick   (bench.py --no-extras, wall clock; 3 alternating rounds; base = $BASE)"
for rep in 1 2 3; do for spec in "$@"; do set -- $spec; for b in base new; do
  lib=""; [ $b = base ] && lib="$LIB"
  v=$(CM3_AMD_LIB=$lib timeout 300 python bench.py --workload $1 --mode $2 --no-extras --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "$1 $2 $b $v"
done; done; done
