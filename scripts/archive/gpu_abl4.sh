#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
W21=$((3+32+256))
V="r6=$((W21)),r6_computeonly=$((W21+(2<<16))),r6_compute_nolds=$((W21+(18<<16)))"
python tools/wide_probe.py --shapes 512x4096x4096 --variants "$V" --iters 40 2>&1 | grep -v amdgpu.ids | cut -c1-110
