#!/bin/bash
# Like pmc_run.sh but for an arbitrary command: tools/pmc_cmd.sh <tag> <command...>
set -u
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  (cd "$R" && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o pmc -- "$@" > "$OUT/pass$i.log" 2>&1)
  echo "pass $i: rc=$?"
done
(cd "$R" && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o st -- "$@" > "$OUT/stats.log" 2>&1)
