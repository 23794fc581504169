#!/bin/bash
# Times tools/mca_bwd_ab.py with each ablation build of the delta pass (tools/mca_delta_ablate_gen.py): the per-call figures differ by the
# delta kernel only.  Usage (GPU box): bash tools/mca_delta_ablate_run.sh > gpurun_out/mca_delta_ablate.txt
cd ${GRAFT_REPO_ROOT:-.}
for n in 0 1 2 3 4 5 6; do
  echo "== variant $n"
  TSCD_HIP_LIB=$PWD/tools/_ablate/libtscd_hip_mca$n.so timeout -k 10 120 python tools/mca_bwd_ab.py 2>&1 | grep -E "P 500 need_reg (True, no class|False round 1)" || exit 1
done
