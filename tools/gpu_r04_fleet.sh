#!/bin/bash
# fleet tick: buckets side by side inside the graph, `given` in the fleet's own buffer
set -o pipefail
mkdir -p gpurun_out/r04f
timeout -k 10 600 python -m pytest tests/test_walkers.py tests/test_gpu_assemble.py -m gpu -x -q -k "fleet or walker or slots or per_cu" > gpurun_out/r04f/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r04f/pytest.txt
timeout -k 10 600 python tools/bench_next_rows.py > gpurun_out/r04f/next_rows.txt 2>&1 && head -5 gpurun_out/r04f/next_rows.txt
