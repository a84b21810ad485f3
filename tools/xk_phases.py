"""Diagnostic: where an exchange-K launch spends its time (needs the QUICK_AMD_TOOLS library: `python -m quick_amd.build --tools`,
QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so).  Per-wave s_memrealtime stamps at the phase boundaries.
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
abl = 16
while args and args[0] in ("--kernel", "--abl", "--env-abl"):
    if args[0] == "--kernel":
        kid = int(args[1], 0)
    elif args[0] == "--env-abl":   # the kernel's ABL value itself (64 = stamps; + 128 no counted wait, 256 no x pieces, 512 no weight loads, 1024 spread pieces)
        os.environ["QUICK_XK_ABL"] = args[1]
    else:
        abl = int(args[1], 0)   # 16 stamps only, 17 loads only, 18 no loads, 19 no loads + no B reads, 20 no exchange, 21 no loads + no dequant,
    args = args[2:]             # 22 MFMAs + barrier, 23 no loads + no barrier, 24 MFMAs only (w4a16_gemm.hip, run_gemm)
DBG = 4096 * 8 * 64
for spec in (args or ["512x4096x4096"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device=dev).half()
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(40)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    plan = kernels.plan_describe(M, K, N, G, kid)
    need = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kid, 0)
    ws = torch.zeros((need + DBG) // 8, dtype=torch.int64, device=dev)
    k16 = kid + (abl << 16)
    REP = 8
    acc, cycs, segs = [], [], []
    for i in range(40 + REP):   # the first 40 warm the clocks up; the rest are read back one by one: HBM-cold weights every time
        qw, sc, qz = sets[i % 40]
        ws[need // 8:].zero_()
        rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, M, K, N, G, k16, 0, None)
        assert rc == 0, _lib.last_error()
        if i >= 40:
            torch.cuda.synchronize()
            raw = ws[need // 8:].cpu().numpy().reshape(-1, 8)
            d = raw[:, :6].astype(np.float64) / 100.0  # us
            keep = d[:, 5] > 0
            acc.append(d[keep])
            cycs.append(raw[keep, 6].astype(np.float64))
            segs.append(raw[keep, 7].astype(np.uint64))
    names = ["entry -> first fragments", "K loop", "K parities swapped through LDS", "slices exchanged (mailboxes)", "way out (image, stores acknowledged)"]
    tot = np.mean([d[:, 5].max() - d[:, 0].min() for d in acc])
    print(f"{spec} abl={abl} env={os.environ.get('QUICK_XK_ABL', '-')}: {plan}\n   {len(acc[0])} waves stamped, {REP} launches; first entry -> last wave done {tot:.2f} us"
          f" (min {min(d[:, 5].max() - d[:, 0].min() for d in acc):.2f})")
    for i, n in enumerate(names):
        v = np.array([[(d[:, i + 1] - d[:, i]).mean(), (d[:, i + 1] - d[:, i]).min(), (d[:, i + 1] - d[:, i]).max()] for d in acc]).mean(0)
        print(f"   {n:40s} mean {v[0]:7.2f} us   min {v[1]:7.2f}   max {v[2]:7.2f}")
    kl = np.mean([(d[:, 2] - d[:, 1]).mean() for d in acc])
    kc = np.mean([c.mean() for c in cycs])
    kl = max(kl, 1e-9)
    print(f"   K loop: {kc:.0f} shader clocks per wave in {kl:.2f} us = {kc / kl / 1000:.3f} GHz")
    loc = np.mean([g.astype(np.float64).mean() for g in segs])
    print(f"   word 7 (XW: share of waves whose tile sat behind one L2): {loc:.3f}")
    sw = np.mean([(g >> np.uint64(32)).astype(np.float64).mean() for g in segs])
    sb = np.mean([(g & np.uint64(0xffffffff)).astype(np.float64).mean() for g in segs])
    if sw + sb > 0:
        print(f"   of those clocks: {sw:.0f} in the counted wait at the end of a stage, {sb:.0f} at the barrier (per wave, all stages)")
    r = np.array([[(d[:, i] - d[:, 0].min()).mean() for i in range(1, 6)] for d in acc]).mean(0)
    print("   phases reached since first entry (mean over waves): " + "  ".join(f"{v:.2f}" for v in r))
