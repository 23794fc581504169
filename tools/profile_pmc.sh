#!/bin/bash
# rocprofv3 PMC evidence of the step's kernel families (north_star: "rocprof HBM GB/s and MFMA utilisation against gfx950 peak").
# Separate passes, each --pmc with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ MFMA / busy cycles, FETCH_SIZE, WRITE_SIZE.
# The program behind `--` is python3 itself.  Usage (GPU box): bash tools/profile_pmc.sh <outdir under gpurun_out> [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; N=${2:-12}
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
grep -c . $OUT/counters_available.txt
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE; do
  grep -q "$C" $OUT/counters_available.txt && echo "have $C" || echo "MISSING $C"
done
timeout -k 10 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o sq -- python3 tools/steady_steps.py $N > $OUT/pmc_sq.log 2>&1 && echo "sq ok" || { tail -5 $OUT/pmc_sq.log; }
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 tools/steady_steps.py $N > $OUT/pmc_fetch.log 2>&1 && echo "fetch ok" || { tail -5 $OUT/pmc_fetch.log; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 tools/steady_steps.py $N > $OUT/pmc_write.log 2>&1 && echo "write ok" || { tail -5 $OUT/pmc_write.log; }
python3 tools/summarise_pmc.py $OUT $N > $OUT/summarise_pmc.log 2>&1; tail -40 $OUT/summarise_pmc.log
rm -rf $OUT/pmc_sq $OUT/pmc_fetch $OUT/pmc_write
