#!/bin/bash
# PMC passes (tools/pmc_run.sh) for the six bench lines of the round, summarised into gpurun_out/pmc_<tag>_summary.txt and
# gpurun_out/pmc_traffic_new.json; the raw counter files are removed on the box (they exceed what gpurun copies back).
set -u
R="${GRAFT_REPO_ROOT:-$PWD}"; cd "$R"
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for spec in "c2_trajectory_n4_e4096 c2 trajectory k_particle_step" "c3_trajectory_n2_e8192 c3 trajectory k_checkers_step_fast" "c5_trajectory_n8_e8192 c5 trajectory k_particle_step" "c2_particle_antipodal_n4_e4096 c2 in-place k_particle_step" "c3_checkers_stage2_n2_e8192 c3 in-place k_checkers_step_fast" "c5_particle_merge8_n8_e8192 c5 in-place k_particle_step"; do
  set -- $spec
  bash tools/pmc_run.sh $1 --workload $2 --mode $3 --no-extras --steps 6 --warmup 2 > gpurun_out/pmc_$1.runlog 2>&1
  cd "$R"
  python tools/pmc_summary.py gpurun_out/pmc_$1 $4 $1 gpurun_out/pmc_traffic_new.json > gpurun_out/pmc_$1_summary.txt 2>&1
  rm -rf gpurun_out/pmc_$1
  echo "== $1"; cat gpurun_out/pmc_$1_summary.txt
done
