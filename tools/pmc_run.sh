#!/bin/bash
# PMC collection on the GPU box: one rocprofv3 pass per counter group (kernel-trace only, as the guide
# prescribes), output under gpurun_out/pmc_<tag>/.  Usage: tools/pmc_run.sh <tag> <bench args...>
set -u
TAG="$1"; shift
R="${GRAFT_REPO_ROOT:-$PWD}"
OUT="$R/gpurun_out/pmc_${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE GRBM_COUNT" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python "$R/bench.py" --no-sweep --no-cpu-baseline "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
find "$OUT" -name "*counter_collection.csv" | head
