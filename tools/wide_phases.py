"""Diagnostic: where a 256-token-tile launch spends its time (ablation value 16 of w4a16_wide_kernel = s_memrealtime stamps
at the phase boundaries of every wave).  python tools/wide_phases.py [MxKxN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
for spec in (sys.argv[1:] or ["256x128x4096", "4096x128x4096", "4096x4096x4096"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device=dev).half()
    qw, sc, qz = packing.random_mi355x(K, N, G, dev)
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    ws = torch.zeros(4096 * 8 * 64 // 8, dtype=torch.int64, device=dev)
    kid = (3 + 128 + 512) + (16 << 16)
    for _ in range(3):
        rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, M, K, N, G, kid, 0, None)
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    nt = -(-M // 256) * (N // 256)
    d = ws.cpu().numpy().reshape(-1, 8)[: nt * 4, :5].astype(np.float64) / 100.0  # us
    t0 = d[:, 0].min()
    names = ["entry -> first stage landed + prepared", "K loop", "epilogue issue", "stores acknowledged"]
    print(f"{spec}: {nt} tiles; wave entry spread {d[:, 0].max() - t0:.2f} us; first entry -> last wave done {d[:, 4].max() - t0:.2f} us")
    for i, n in enumerate(names):
        v = d[:, i + 1] - d[:, i]
        print(f"   {n:42s} mean {v.mean():7.2f} us   min {v.min():7.2f}   max {v.max():7.2f}")
