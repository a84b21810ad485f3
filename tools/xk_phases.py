"""Diagnostic: where an exchange-K launch spends its time (needs the QUICK_AMD_TOOLS library: `python -m quick_amd.build --tools`,
QUICK_AMD_LIB_OVERRIDE=quick_amd/lib/libquick_amd_tools.so).  Per-wave s_memrealtime stamps at the phase boundaries.
    python tools/xk_phases.py [--kernel ID] [MxKxN ...]        ID: QUICK_KERNEL_XK | mb << 4 | slices << 8 ..., default 128-token tiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
args = sys.argv[1:]
kid = 4 | (4 << 4)
if args and args[0] == "--kernel":
    kid = int(args[1], 0)
    args = args[2:]
DBG = 4096 * 8 * 64
for spec in (args or ["512x4096x4096"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device=dev).half()
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(40)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    plan = kernels.plan_describe(M, K, N, G, kid)
    need = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kid, 0)
    ws = torch.zeros((need + DBG) // 8, dtype=torch.int64, device=dev)
    k16 = kid + (16 << 16)
    for i in range(40):   # the last launch is the one read back: HBM-cold weights, warm clocks
        qw, sc, qz = sets[i]
        ws[need // 8:].zero_()
        rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, M, K, N, G, k16, 0, None)
        assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    d = ws[need // 8:].cpu().numpy().reshape(-1, 8)[:, :6].astype(np.float64) / 100.0  # us
    d = d[d[:, 5] > 0]
    t0 = d[:, 0].min()
    names = ["entry -> stages 0, 1 landed + first fragments", "K loop", "K parities swapped through LDS", "slices exchanged (mailboxes)",
             "way out (image, stores acknowledged)"]
    print(f"{spec}: {plan}\n   {len(d)} waves stamped; wave entry spread {d[:, 0].max() - t0:.2f} us; first entry -> last wave done {d[:, 5].max() - t0:.2f} us")
    for i, n in enumerate(names):
        v = d[:, i + 1] - d[:, i]
        print(f"   {n:50s} mean {v.mean():7.2f} us   min {v.min():7.2f}   max {v.max():7.2f}")
    for i in range(1, 6):
        print(f"   phase {i} reached (since first entry): mean {(d[:, i] - t0).mean():6.2f}  min {(d[:, i] - t0).min():6.2f}  max {(d[:, i] - t0).max():6.2f} us")
