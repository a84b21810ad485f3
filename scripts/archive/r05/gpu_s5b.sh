#!/bin/bash
# A/B of the 256 x 256 tile's exit modes on prefill shapes (tools/ab_xw_exit.py), then the stream test that failed in s5a
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/s5b; mkdir -p $out
timeout 900 python tools/ab_xw_exit.py 2>&1 | grep -v amdgpu.ids | tee $out/ab_xw_exit.txt
timeout 600 python -m pytest tests -q -m gpu -x -k "other_streams or forced_four_wave or 256x256" 2>&1 | tail -3
