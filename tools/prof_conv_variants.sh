cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02e
mkdir -p $O
cd $R
for V in 3 4 5; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmcA_v$V -- python3 tools/conv_one.py --variant $V --iters 10 > $O/pmcA_v$V.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmcB_v$V -- python3 tools/conv_one.py --variant $V --iters 10 > $O/pmcB_v$V.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 tools/conv_one.py --variant 5 --iters 10 > $O/trace.log 2>&1
ls $O
