#!/bin/bash
# product loop: parity tests of the family, then the audit of every XM selection against the other families' pick (QUICK_AMD_XM=0) on the layer shapes
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "xm" -x > gpurun_out/r06/pytest_xm.txt 2>&1; tail -2 gpurun_out/r06/pytest_xm.txt
QUICK_AMD_XM=0 timeout 1500 python tools/xm_audit.py > gpurun_out/r06/xm_audit.txt 2>&1; tail -3 gpurun_out/r06/xm_audit.txt
