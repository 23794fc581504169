#!/bin/bash
# rocprofv3 kernel trace of tools/steady_steps.py -> per-step kernel statistics and the post-phase breakdown (what follows the assignment chain).
# Usage (on the GPU box): bash tools/profile_post.sh <outdir under gpurun_out> [steps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; N=${2:-120}
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/steady -o steady -- python3 tools/steady_steps.py $N > $OUT/steady_run.txt 2>&1 || { tail -5 $OUT/steady_run.txt; exit 1; }
T=$(find $OUT/steady -name '*kernel_trace.csv' | head -1); S=$(find $OUT/steady -name '*kernel_stats.csv' | head -1)
python3 tools/kernel_stats_per_step.py $S $N > $OUT/steady_kernel_stats.txt 2>&1
python3 tools/post_phase_kernels.py $T 40 ${SEQ:+--seq} > $OUT/post_phase_kernels.txt 2>&1   # SEQ=1: plus the launch sequence of the last step
rm -rf $OUT/steady
tail -2 $OUT/steady_run.txt; head -60 $OUT/post_phase_kernels.txt
