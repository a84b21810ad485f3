#!/bin/bash
# full GPU parity suite + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/tests
timeout 1700 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/tests/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/tests/pytest_gpu.log
tail -30 gpurun_out/tests/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
