#!/bin/bash
# one GPU-box call: full GPU test suite, smoke(), the official bench line, rocprof kernel stats of the same command, PMC passes
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 400 python -m pytest tests/ -q -m gpu --tb=short > gpurun_out/final/tests.log 2>&1; echo "tests rc=$?"; tail -1 gpurun_out/final/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
timeout -k 10 500 python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/final/pmc_write.log 2>&1; echo "write rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/final/bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "roof", d["roofline"]["achieved"], d["roofline"]["dominant_shape"]["achieved"])
print("parity", json.dumps(d.get("parity_full_size"))[:600])
PY
