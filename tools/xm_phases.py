"""Diagnostic: anatomy of a mid-token launch (needs the QUICK_AMD_TOOLS library: `python -m quick_amd.build --tools`,
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so).  Per-wave s_memrealtime stamps (10 ns ticks), HBM-cold weights.
    python tools/xm_phases.py [--pr 1,2,3] [MxKxN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
args = sys.argv[1:]
prs = [1, 2, 3]
extra = 0
if args and args[0] == "--t32":
    extra = 1 << 8
    args = args[1:]
if args and args[0] == "--pr":
    prs = [int(v) for v in args[1].split(",")]
    args = args[2:]
DBG = 4096 * 8 * 64
NAMES = ["entry", "x(0, 0) landed", "W(0) landed", "end of stage 0", "end of stage 1", "end of stage 2", "end of stage 3", "loop left", "exit (stores acknowledged)"]
for spec in (args or ["64x4096x4096", "64x4096x22016"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    nsets = max(2, min(40, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    for pr in prs:
        kid = 7 | (pr << 4) | extra
        plan = kernels.plan_describe(M, K, N, G, kid)
        ws = torch.zeros((DBG + (1 << 20)) // 8, dtype=torch.int64, device=dev)
        k16 = kid + (16 << 16)
        REP = 8
        acc, clk = [], []
        for i in range(nsets + REP):
            qw, sc, qz = sets[i % nsets]
            ws.zero_()
            rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(), ws.numel() * 8, M, K, N, G, k16, 0, None)
            assert rc == 0, _lib.last_error()
            if i >= nsets:
                torch.cuda.synchronize()
                raw = ws.cpu().numpy()[-(DBG // 8):].reshape(-1, 8).astype(np.uint64)
                raw = raw[raw[:, 0] > 0]
                ent = raw[:, 0]
                cols = [ent]
                for j in (1, 2, 3):
                    for half in (0, 1):
                        lo = (raw[:, j] >> np.uint64(32 * half)) & np.uint64(0xffffffff)
                        full = (ent & ~np.uint64(0xffffffff)) | lo
                        full = np.where(full < ent, full + np.uint64(1 << 32), full)
                        full = np.where(lo == 0, ent, full)      # (a stage the wave does not have)
                        cols.append(full)
                cols += [raw[:, 4], raw[:, 5]]
                acc.append(np.stack(cols, 1).astype(np.float64) / 100.0)
                # core clock of the wave between entry and the end of its loop: s_memtime ticks over the 100 MHz stamps
                dt = (raw[:, 4] - raw[:, 0]).astype(np.float64) / 100.0
                dc = (raw[:, 7] - raw[:, 6]).astype(np.float64)
                clk.append(float(np.median(dc[dt > 0] / dt[dt > 0]) / 1e3))
        tot = np.mean([d[:, 8].max() - d[:, 0].min() for d in acc])
        print(f"{spec} pr={pr}: {plan}\n   {len(acc[0])} waves stamped, {REP} launches; first entry -> last exit {tot:.2f} us; core clock entry -> loop left (median over waves) {np.mean(clk):.2f} GHz")
        print(f"   {'phase (us since the first wave entered)':42s} {'first':>7s} {'mean':>7s} {'last':>7s}    own: min  mean  max (since the wave's previous stamp)")
        for i, n in enumerate(NAMES):
            rel = [d[:, i] - d[:, 0].min() for d in acc]
            own = [d[:, i] - d[:, max(i - 1, 0)] for d in acc]
            print(f"   {n:42s} {np.mean([r.min() for r in rel]):7.2f} {np.mean([r.mean() for r in rel]):7.2f} {np.mean([r.max() for r in rel]):7.2f}"
                  f"         {np.mean([o.min() for o in own]):5.2f} {np.mean([o.mean() for o in own]):5.2f} {np.mean([o.max() for o in own]):5.2f}")
