#!/bin/bash
# One GPU-box call for the round-4 evidence -> gpurun_out/final/summary/ (copied into profiles/ afterwards):
#   bench line + rocprofv3 kernel stats of the bench command + the conv family's two PMC traffic passes (tools/profile_bench.sh, summarise_profiles.py),
#   steady-state per-step kernel table + post-phase breakdown (tools/profile_post.sh), step timelines (plain / ArenaDDP with one rank),
#   PMC per kernel family: MFMA busy, fabric bandwidth (tools/profile_pmc.sh).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final/summary
bash tools/profile_bench.sh || exit 1
python3 tools/summarise_profiles.py r04 gpurun_out/final/summary > gpurun_out/final/summarise.log 2>&1 || { tail -5 gpurun_out/final/summarise.log; exit 1; }
rm -rf gpurun_out/final/stats gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write
bash tools/profile_post.sh final/post 150 > gpurun_out/final/post.log 2>&1 || { tail -5 gpurun_out/final/post.log; exit 1; }
cp gpurun_out/final/post/steady_kernel_stats.txt gpurun_out/final/summary/r04_steady_kernel_stats_final.txt
cp gpurun_out/final/post/post_phase_kernels.txt gpurun_out/final/summary/r04_post_phase_kernels_final.txt
timeout -k 10 200 python tools/step_timeline.py > gpurun_out/final/summary/r04_step_timeline_final.txt 2>&1 || exit 1
TSCD_TIMELINE_DDP=1 timeout -k 10 200 python tools/step_timeline.py > gpurun_out/final/summary/r04_step_timeline_ddp_final.txt 2>&1 || exit 1
bash tools/profile_pmc.sh final/pmc 10 > gpurun_out/final/pmc.log 2>&1 || { tail -5 gpurun_out/final/pmc.log; exit 1; }
cp gpurun_out/final/pmc/pmc_mfma.json gpurun_out/final/summary/r04_pmc_mfma.json
cp gpurun_out/final/pmc/pmc_hbm.json gpurun_out/final/summary/r04_pmc_hbm.json
grep steady gpurun_out/final/summary/r04_step_timeline_final.txt gpurun_out/final/summary/r04_step_timeline_ddp_final.txt
head -c 700 gpurun_out/final/bench_line.json; echo; tail -16 gpurun_out/final/pmc.log
bash tools/lap_ab/run.sh > gpurun_out/final/summary/r04_lap_ns_per_pass.txt 2>&1; tail -3 gpurun_out/final/summary/r04_lap_ns_per_pass.txt
timeout -k 10 120 python tools/mca_bwd_ab.py > gpurun_out/final/summary/r04_mca_call_times.txt 2>&1; grep "P 500" gpurun_out/final/summary/r04_mca_call_times.txt
