#!/bin/bash
# One GPU-box call for the round-3 evidence: bench line + rocprofv3 kernel stats of the bench + the two PMC traffic passes (tools/profile_bench.sh),
# then the steady-state per-step kernel statistics and the post-phase breakdown (rocprofv3 kernel trace of tools/steady_steps.py), the step
# timeline, the per-shape conv table and the assignment drift.  tools/summarise_profiles.py r03 + the copies at the end of this script's
# caller turn gpurun_out/final/* into profiles/r03_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
bash tools/profile_bench.sh || exit 1
python3 tools/summarise_profiles.py r03 gpurun_out/final/summary > gpurun_out/final/summarise.log 2>&1 || { tail -5 gpurun_out/final/summarise.log; exit 1; }
rm -rf gpurun_out/final/stats gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write      # raw rocprofv3 output: too large to carry back
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/steady -o steady -- python3 tools/steady_steps.py 150 > gpurun_out/final/steady_run.txt 2>&1 || exit 1
T=$(find gpurun_out/final/steady -name '*kernel_trace.csv' | head -1); S=$(find gpurun_out/final/steady -name '*kernel_stats.csv' | head -1)
python3 tools/kernel_stats_per_step.py $S 150 > gpurun_out/final/steady_kernel_stats.txt 2>&1
python3 tools/post_phase_kernels.py $T 40 > gpurun_out/final/post_phase_kernels.txt 2>&1
rm -rf gpurun_out/final/steady
timeout -k 10 200 python tools/step_timeline.py > gpurun_out/final/step_timeline.txt 2>&1 || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --conv-table > gpurun_out/final/bench_convtable.json 2> gpurun_out/final/conv_table.txt || exit 1
timeout -k 10 300 python tools/lap_drift.py > gpurun_out/final/lap_drift.txt 2>&1 || exit 1
tail -3 gpurun_out/final/steady_run.txt; grep steady gpurun_out/final/step_timeline.txt; tail -3 gpurun_out/final/lap_drift.txt; cat gpurun_out/final/bench_line.json | head -c 600
