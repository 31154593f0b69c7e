#!/bin/bash
# Alternates bench.py between library builds on ONE box: tools/ab_libs.sh "<workload> <mode>" <lib name or - for the product> ...
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"; cd "$R"
SPEC="$1"; shift
set -- $SPEC "$@"; WL="$1"; MODE="$2"; shift 2
for rep in 1 2 3; do for b in "$@"; do
  lib=""; [ "$b" != "-" ] && lib="$R/tools/variants/libcm3_hip_$b.so"
  v=$(CM3_AMD_LIB=$lib timeout 300 python bench.py --workload $WL --mode $MODE --no-extras --no-sweep --no-cpu-baseline --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.readline())['us_per_tick'])")
  echo "$WL $MODE $b $v"
done; done
