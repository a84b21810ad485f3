"""Diagnostic: the four-tile fragment flavour of the skinny kernel against the oracle, per 16-channel tile, under whatever library
QUICK_AMD_LIB_OVERRIDE names (tools/bin/ab_uncond*.so: the chunk loop with unconditional requests).    python tools/skinny_uncond_check.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from quick_amd import kernels
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
SK4 = 1 | (4 << 4)
bad_total = 0
for (M, K, N) in ((16, 1024, 256), (16, 2048, 1024), (5, 1024, 128), (16, 8192, 1024), (9, 4096, 2048), (33, 1024, 256), (64, 4096, 512), (48, 2048, 1024)):
    x, iw, s, z = oracle.make_synthetic(M, K, N, 128, seed=M + K + N)
    want = oracle.w4a16_forward(x, iw, s, z, 128).astype(np.float32)
    packed = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in oracle.pack_mi355x(iw, s, z))
    xd = torch.from_numpy(x).to(dev)
    plan = kernels.plan_describe(M, K, N, 128, SK4)
    for r in range(reps):
        y = kernels.gemm_forward(xd, *packed, kernel_id=SK4).float().cpu().numpy()
        d = np.abs(np.nan_to_num(y, nan=1e9, posinf=1e9, neginf=1e9) - want)
        tiles = np.where(d.reshape(M, N // 16, 16).max((0, 2)) > 2e-3 * np.abs(want).max())[0]
        bad_total += len(tiles) > 0
        print(f"{M:3d} x {K} x {N} [{plan.split(' grid')[0]}] run {r}: nonfinite {int((~np.isfinite(y)).sum())}  bad tiles {tiles[:24].tolist()}{' ...' if len(tiles) > 24 else ''}", flush=True)
print("launches with wrong tiles:", bad_total)
