"""Diagnostic: where a wide / ring launch spends its time (ablation value 16 = s_memrealtime stamps at the phase boundaries of
every wave: 256 x 256 double-buffered kernel, 64 x 128 and 128 x 128 eight-wave ring kernels).
    python tools/wide_phases.py [--kernel ID] [MxKxN ...]        ID: as tools/wide_probe.py variants, default the 256 x 256 tile"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
args = sys.argv[1:]
kid = 3 + 128 + 512
if args and args[0] == "--kernel":
    kid = int(args[1], 0)
    args = args[2:]
DBG = 4096 * 8 * 64
for spec in (args or ["256x128x4096", "4096x128x4096", "4096x4096x4096"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device=dev).half()
    qw, sc, qz = packing.random_mi355x(K, N, G, dev)
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    plan = kernels.plan_describe(M, K, N, G, kid)
    need = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kid, 0)
    ws = torch.zeros((need + DBG) // 8, dtype=torch.int64, device=dev)
    k16 = kid + (16 << 16)
    for _ in range(3):
        ws[need // 8:].zero_()
        rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, M, K, N, G, k16, 0, None)
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    d = ws[need // 8:].cpu().numpy().reshape(-1, 8)[:, :5].astype(np.float64) / 100.0  # us
    d = d[d[:, 4] > 0]                                                                  # (K split: only the finishing workgroups stamp)
    t0 = d[:, 0].min()
    names = ["entry -> first stage landed + prepared", "K loop", "K halves added (eight waves)" if "waves=8" in plan else "epilogue issue",
             "way out (K-split reduction, LDS image, stores)" if "waves=8" in plan else "stores acknowledged"]
    print(f"{spec}: {plan[:86]}\n   {len(d)} waves stamped; wave entry spread {d[:, 0].max() - t0:.2f} us; first entry -> last wave done {d[:, 4].max() - t0:.2f} us")
    for i, n in enumerate(names):
        v = d[:, i + 1] - d[:, i]
        print(f"   {n:48s} mean {v.mean():7.2f} us   min {v.min():7.2f}   max {v.max():7.2f}")
