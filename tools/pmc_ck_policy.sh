#!/bin/bash
# rocprofv3 evidence for the one-launch Checkers policy rollout at C3 (tools/ck_policy_worker.py: 3 + 20 collects of 33 ticks):
# (1) kernel-trace + stats (the kernel is ~450 us long: the profiler's per-dispatch stretch, which ruins the 2 us step launches,
# is < 1 % here), (2) one pass per PMC group, kernel-trace only.  Run on the GPU box:  gpurun -- 'bash tools/pmc_ck_policy.sh'
#   -> gpurun_out/r06_ck_policy_kernel_stats.txt, gpurun_out/r06_pmc_ck_policy_summary.txt
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; OUT="$R/gpurun_out/pmc_ck_policy"; mkdir -p "$OUT"
python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.build()" >/dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 python "$R/tools/ck_policy_worker.py" 20 > "$OUT/unprofiled.log" 2>&1; echo "unprofiled: $(tail -1 $OUT/unprofiled.log)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o ckp -- python "$R/tools/ck_policy_worker.py" 20 > "$OUT/stats.log" 2>&1
echo "stats rc=$? $(tail -1 $OUT/stats.log | cut -c1-100)"
cd "$R"; python tools/rocprof_summary.py "$OUT/stats" > gpurun_out/r06_ck_policy_kernel_stats.txt 2>&1
{ echo; echo "unprofiled run of the same command: $(tail -1 $OUT/unprofiled.log)"; echo "under rocprofv3 --kernel-trace --stats: $(tail -1 $OUT/stats.log)"; } >> gpurun_out/r06_ck_policy_kernel_stats.txt
head -12 gpurun_out/r06_ck_policy_kernel_stats.txt
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pass$i" -o pmc -- \
      python "$R/tools/ck_policy_worker.py" 20 > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$? $(tail -1 $OUT/pass$i.log | cut -c1-80)"
done
cd "$R"
python tools/pmc_summary.py "$OUT" k_ck_policy_rollout > gpurun_out/r06_pmc_ck_policy_summary.txt 2>&1
rm -rf "$OUT"
cat gpurun_out/r06_pmc_ck_policy_summary.txt
