#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
V="tiled=2,w4x2nr=$((3+64+512+4096)),w8x2=$((3+128+512)),w2x1nr=$((3+32+256+4096))"
for fill in randn zeros ones; do
  echo "### xfill=$fill"
  python tools/wide_probe.py --shapes 512x4096x4096,4096x8192x8192 --variants "$V" --xfill $fill --iters 30 2>&1 | grep -v amdgpu.ids | cut -c1-100
done
