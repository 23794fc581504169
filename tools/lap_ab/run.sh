#!/bin/bash
# ns per relaxation pass of the host assignment solver on this machine's CPU (no GPU work): 8 matrices of 500 x 500 bootstrapped from the
# device cost sample in tests/golden (random-init regime: ~70 passes per row, 8 tied columns per pass), results checked against scipy.
# Usage: bash tools/lap_ab/run.sh            (round 4 on the pool's EPYC 9575F: 55.1 ns before the plateau pass, 51.4 ns with it)
cd ${GRAFT_REPO_ROOT:-.} && python3 tools/lap_ab/gen.py || exit 1
lscpu | grep -E "Model name" | head -1
g++ -O3 -std=c++17 -fno-fast-math -pthread tools/lap_ab/bench.cpp -o /tmp/lapb_new 2>/dev/null || { echo build failed; exit 1; }
for rep in 1 2 3; do echo "$(taskset -c 4 /tmp/lapb_new 4 | awk '{print $(NF-3)}' | sort -n | head -1) ns/pass"; done
