#!/bin/bash
# first GPU pass for the wide kernel: parity subset, then the timing probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/wide1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gemm_gpu.py -x -q -m gpu -k "synthetic_sweep or golden_fixture_forward" > gpurun_out/wide1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/wide1/pytest.log
tail -5 gpurun_out/wide1/pytest.log
timeout 900 python tools/wide_probe.py --out gpurun_out/wide1/probe.jsonl > gpurun_out/wide1/probe.log 2>&1
echo "probe rc=$?" >> gpurun_out/wide1/probe.log
cat gpurun_out/wide1/probe.log
