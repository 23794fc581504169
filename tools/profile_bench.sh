#!/bin/bash
# One GPU-box call: rocprofv3 kernel stats of the bench command + the two PMC traffic passes (separate, as the guide prescribes).
# Outputs under gpurun_out/final/; tools/summarise_profiles.py <tag> then writes the committed summaries under profiles/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err && echo "bench ok" &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/stats.log 2>&1 && echo "stats ok"
if [ "$1" != "stats-only" ]; then
set -e
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > gpurun_out/final/pmc_fetch.log 2>&1 && echo "fetch ok"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --steps 2 --warmup 1 --preroll 0 --no-cpu-baseline > gpurun_out/final/pmc_write.log 2>&1 && echo "write ok"
fi
