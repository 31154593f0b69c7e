#!/bin/bash
# PMC passes for the whole-episode policy rollout kernel at C2 (tools/policy_row_tiles.py --worker: 3 + 60 launches of
# cm3_policy_rollout_f32, 33 ticks each): one rocprofv3 pass per counter group, kernel-trace only.  Run on the GPU box:
#   gpurun -- 'bash tools/pmc_policy.sh'        -> gpurun_out/pmc_policy_c2_summary.txt
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$R/gpurun_out/pmc_policy_c2"; mkdir -p "$OUT"
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python "$R/tools/policy_row_tiles.py" --worker particle_stage2_antipodal 4 4096 f16x3 > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$? $(tail -1 $OUT/pass$i.log | cut -c1-60)"
done
cd "$R"
python tools/pmc_summary.py "$OUT" k_policy_rollout > gpurun_out/pmc_policy_c2_summary.txt 2>&1
rm -rf "$OUT"
cat gpurun_out/pmc_policy_c2_summary.txt
