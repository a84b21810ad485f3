"""Diagnostic: anatomy of a lean small-M launch (needs the QUICK_AMD_TOOLS library: `python -m quick_amd.build --tools`,
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so).  Per-wave s_memrealtime stamps (10 ns ticks), HBM-cold weights.
    python tools/lean_phases.py [--waves 4|8|16] [MxKxN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
args = sys.argv[1:]
waves_list = [4, 8]
use_ln = "--ln" in args
args = [a for a in args if a != "--ln"]
if args and args[0] == "--waves":
    waves_list = [int(v) for v in args[1].split(",")]
    args = args[2:]
DBG = 4096 * 8 * 64
NAMES = ["entry", "indices known (wave, block, k range)", "descriptors built", "x DMA issued", "all requests out (group words, weight tiles)", "x landed in LDS", "unit sums (MFMA), group words gathered", "first weight tile landed",
         "last weight tile landed", "last tile multiplied", "partial in LDS", "barrier passed", "stores acknowledged (wave 0) / exit"]
for spec in (args or ["1x4096x4096", "8x4096x4096", "1x4096x22016", "8x4096x22016"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    nsets = max(2, min(40, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    lnw = (torch.rand(K, device=dev) + 0.5).half()
    for waves in waves_list:
        kid = 6 | ((waves // 4) << 8)
        plan = kernels.plan_describe(M, K, N, G, kid)
        if "tiles_per_wave<=0" in plan:
            continue
        ws = torch.zeros(DBG // 8, dtype=torch.int64, device=dev)
        k16 = kid + (16 << 16)
        REP = 10
        acc = []
        for i in range(nsets + REP):
            qw, sc, qz = sets[i % nsets]
            ws.zero_()
            if use_ln:
                fu = _lib.GemmFusion(None, None, lnw.data_ptr(), 1e-5, 0)
                import ctypes
                rc = lib.quick_w4a16_gemm_f16_fused(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), ctypes.byref(fu), y.data_ptr(), ws.data_ptr(),
                                                    ws.numel() * 8, M, K, N, G, k16, 0, None)
            else:
                rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                                 ws.numel() * 8, M, K, N, G, k16, 0, None)
            assert rc == 0, _lib.last_error()
            if i >= nsets:
                torch.cuda.synchronize()
                raw = ws.cpu().numpy().reshape(-1, 16).astype(np.float64) / 100.0
                raw = raw[raw[:, 9] > 0]
                acc.append(np.concatenate([raw[:, :1], raw[:, 11:13], raw[:, 10:11], raw[:, 1:10]], axis=1))
        nw = len(acc[0])
        tot = np.mean([d[:, 12].max() - d[:, 0].min() for d in acc])
        print(f"{spec}{' +rmsnorm' if use_ln else ''}: {plan}\n   {nw} waves stamped (of {N // 16 * waves}), {REP} launches; first entry -> last exit {tot:.2f} us")
        print(f"   {'phase (us since the first wave entered)':52s} {'first':>7s} {'mean':>7s} {'last':>7s}    own: min  mean  max (since the wave's previous stamp)")
        for i, n in enumerate(NAMES):
            rel = [d[:, i] - d[:, 0].min() for d in acc]
            own = [d[:, i] - d[:, max(i - 1, 0)] for d in acc]
            print(f"   {n:52s} {np.mean([r.min() for r in rel]):7.2f} {np.mean([r.mean() for r in rel]):7.2f} {np.mean([r.max() for r in rel]):7.2f}"
                  f"         {np.mean([o.min() for o in own]):5.2f} {np.mean([o.mean() for o in own]):5.2f} {np.mean([o.max() for o in own]):5.2f}")
        # entry -> first request by launch order of the wave within its workgroup's CU: does the first wave pay for the instruction cache?
        d = acc[-1]
        order = np.argsort(d[:, 0])
        own1 = (d[:, 3] - d[:, 0])[order]
        n = len(own1)
        print("   entry -> x DMA issued by entry order (eighths of the waves): " + "  ".join(f"{own1[i * n // 8:(i + 1) * n // 8].mean():.2f}" for i in range(8)))
