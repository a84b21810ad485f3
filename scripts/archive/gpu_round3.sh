#!/bin/bash
# round-3 evidence: full GPU test suite, bench line, rocprofv3 kernel stats of the same command
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r03_tests.txt
timeout 900 python bench.py --steps 20 --warmup 5 2> gpurun_out/r03_bench.log | tail -1 > gpurun_out/r03_bench.json
tail -30 gpurun_out/r03_bench.log
