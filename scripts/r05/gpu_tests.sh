#!/bin/bash
# full GPU parity suite + smoke (+ a default bench line)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 2700 python -m pytest tests -q -m gpu -x --durations=10 > gpurun_out/r05/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log
tail -30 gpurun_out/r05/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
tail -c 6000 gpurun_out/r05/bench_default.err
