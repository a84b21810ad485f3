#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests -q -m gpu -x -k "workspace_check" 2>&1 | tail -3
timeout 300 python tools/time_ops.py 64 2>/dev/null | tee gpurun_out/r05/time_ops_b64.txt
python - <<'PY'
from quick_amd import kernels
for s in [(64,4096,12288),(64,4096,4096),(64,4096,22016),(64,11008,4096),(64,4096,6144),(64,4096,28672),(64,14336,4096)]:
    print(s, kernels.plan_describe(*s,128))
PY
