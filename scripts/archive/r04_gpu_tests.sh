#!/bin/bash
# full GPU parity suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r04/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04/pytest_gpu.log
tail -40 gpurun_out/r04/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
