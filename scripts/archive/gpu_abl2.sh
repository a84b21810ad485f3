#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256))
V="r6_loadonly=$((W21+(1<<16))),r6_wonly=$((W21+(5<<16))),r6_xonly=$((W21+(9<<16)))"
python tools/wide_probe.py --shapes 512x4096x4096,2048x4096x4096,512x8192x8192 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-110
