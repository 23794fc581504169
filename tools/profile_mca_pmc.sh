#!/bin/bash
# SQ counters of the MCA kernels alone at the bench clip's sizes (tools/mca_bwd_ab.py): wave / wait / instruction-mix counters in three passes.
# Usage (GPU box): bash tools/profile_mca_pmc.sh <outdir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA"
P3="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 tools/mca_bwd_ab.py > $OUT/p$i.log 2>&1 && echo "pass $i ok" || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 tools/pmc_kernels.py "mca_" $OUT/p1 $OUT/p2 $OUT/p3 > $OUT/mca_pmc.txt 2>&1
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
grep -A26 "mca_aggregate_kernel<64, true\|mca_bwd_delta_kernel<64, true" $OUT/mca_pmc.txt | head -120
