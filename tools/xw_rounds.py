"""Diagnostic: what a workgroup of a MULTI-ROUND `xw` 256 x 256 launch spends outside its K loop (what persistent workgroups
with an overlapped way out could win).  Needs the tools library (QUICK_AMD_LIB_OVERRIDE=tools/bin/libquick_amd_tools.so).  Per-wave
s_memrealtime stamps (w4a16_xw.hpp, ABL & 64): 0 entry, 1 loop begins (behind the prologue's barrier), 2 loop ends, 3 = 4 rows begin (S = 1),
4 rows issued, 5 stores acknowledged.
    python tools/xw_rounds.py [MxKxN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
kid = 5 | (8 << 4) | (1 << 8)          # XW, 256 x 256, one slice
DBG = 4096 * 8 * 64
for spec in (sys.argv[1:] or ["8192x4096x22016"]):
    M, K, N = (int(v) for v in spec.split("x"))
    x = torch.randn(M, K, device=dev).half()
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(3)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    need = lib.quick_w4a16_workspace_bytes_ex(M, K, N, G, kid, 0)
    ws = torch.zeros((need + DBG) // 8, dtype=torch.int64, device=dev)
    k16 = kid + (16 << 16)
    print(f"{spec}: {kernels.plan_describe(M, K, N, G, kid)}")
    for i in range(6):
        qw, sc, qz = sets[i % 3]
        ws[need // 8:].zero_()
        rc = lib.quick_w4a16_gemm_f16_ex(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), None, y.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 8, M, K, N, G, k16, 0, None)
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        if i < 3:
            continue
        raw = ws[need // 8:].cpu().numpy().reshape(-1, 8)
        d = raw[:, :6].astype(np.float64) / 100.0   # us
        d = d[d[:, 5] > 0].reshape(-1, 4, 6)          # [workgroup][wave][stamp]
        wg0, wg1 = d[:, :, 0].min(1), d[:, :, 5].max(1)   # first wave in, last wave's stores acknowledged
        lo0, lo1 = d[:, :, 1].max(1), d[:, :, 2].max(1)
        span = wg1.max() - wg0.min()
        nwg = len(d)
        ncu = 256
        head, loop, out = lo0 - wg0, lo1 - lo0, wg1 - lo1
        rows = (d[:, :, 4] - d[:, :, 3]).max(1)
        busy = (wg1 - wg0).sum()
        # greedy reconstruction of the per-CU chains: a workgroup that starts takes the place of the one that ended last before it
        order = np.argsort(wg0)
        ends = sorted(wg1[order[:ncu]].tolist())
        gaps = []
        import bisect
        for j in order[ncu:]:
            k = bisect.bisect_right(ends, wg0[j]) - 1
            if k < 0: continue
            gaps.append(wg0[j] - ends[k]); ends.pop(k); bisect.insort(ends, wg1[j])
        gaps = np.array(gaps) if gaps else np.zeros(1)
        last_round = (wg0 > wg0.min() + span - 1.02 * np.median(wg1 - wg0)).sum()
        print(f"  launch {i}: {nwg} workgroups, first entry -> last acknowledged {span:8.2f} us; per workgroup: head {head.mean():5.2f} (min {head.min():.2f} max {head.max():.2f})"
              f"  loop {loop.mean():7.2f} (min {loop.min():.2f} max {loop.max():.2f})  way out {out.mean():5.2f} (min {out.min():.2f} max {out.max():.2f}; rows issued in {rows.mean():.2f})")
        print(f"            in-workgroup time / (256 CUs x span) = {busy / (ncu * span):.4f}; loop share {loop.sum() / (ncu * span):.4f}; head share {head.sum() / (ncu * span):.4f};"
              f" way-out share {out.sum() / (ncu * span):.4f}; gap between a workgroup's end and its successor's entry: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}"
              f" ({len(gaps)} successions = {gaps.sum() / (ncu * span):.4f} of CU time); workgroups that entered in the last workgroup-time: {last_round}")
