#!/bin/bash
# Builds tools/variants/libcm3_hip_base.so from a commit's sources (same-box A/B of two builds; CM3_AMD_LIB selects it).  Measurement
# artefact: remove it afterwards.
set -eu
R="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"; cd "$R"
mkdir -p "$R/tools/variants"
BASE="${1:-HEAD}"
T=$(mktemp -d); mkdir -p "$T/csrc" "$T/include"
for f in $(git ls-tree --name-only "$BASE" cm3_amd/csrc/); do git show "$BASE:$f" > "$T/csrc/$(basename "$f")"; done
git show "$BASE:include/cm3_amd.h" > "$T/include/cm3_amd.h"
sed -i 's#"../../include/cm3_amd.h"#"../include/cm3_amd.h"#' "$T/csrc/common.h" "$T/csrc/build.sh"
CM3_SKIP_ISA_LINT=1 CM3_OUT="$R/tools/variants/libcm3_hip_${BASE_NAME:-base}.so" bash "$T/csrc/build.sh"
rm -rf "$T"
