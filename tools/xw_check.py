"""Bring-up check of the 128 x 256 four-wave kernels (QUICK_KERNEL_XW): results against a dense fp32 product of the GPU-dequantised
weights, bit-equality of the one-slice launch with the r02 wide<4, 2> kernel (same summation order), and run-to-run identity.
    python tools/xw_check.py [MxKxN ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = int(os.environ.get("G", "128"))
XW = 5
shapes = sys.argv[1:] or ["128x512x256", "128x1024x256", "130x1024x512", "300x2048x512", "512x4096x4096", "77x4096x256", "1x512x256", "1000x1024x768", "256x256x256", "2049x384x512"]
bad = 0
for spec in shapes:
    M, K, N = (int(v) for v in spec.split("x"))
    torch.manual_seed(M + K + N)
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    qw, sc, qz = packing.random_mi355x(K, N, G, dev)
    w = torch.empty(K, N, dtype=torch.float16, device=dev)
    rc = lib.quick_dequantize_mi355x_f16(qw.data_ptr(), sc.data_ptr(), qz.data_ptr(), w.data_ptr(), K, N, G, None)
    assert rc == 0
    ref = x.float() @ w.float()
    scale = ref.abs().max().item()
    for mb, pairs in ((8, 2), (4, 2), (4, 1), (2, 1)):
      if N % (pairs * 128):
        continue
      wide = kernels.gemm_forward(x, qw, sc, qz, kernel_id=3 | (mb << 4) | (pairs << 8) | (1 << 12), grid_split_k=1)
      for s in (1, 2, 4):
        kid = XW | (s << 8) | ((mb << 4) if mb != 4 else 0) | ((1 << 12) if pairs == 1 else 0)
        plan = kernels.plan_describe(M, K, N, G, kid)
        if f"slices={s}" not in plan or not plan.startswith("xw"):
            continue
        kid |= int(os.environ.get("XW_EXTRA_BITS", "0"), 0)   # (tools library: 0x10000 = the stamped build, with QUICK_XW_EXP=64 the early-barrier loop)
        try:
            y = kernels.gemm_forward(x, qw, sc, qz, kernel_id=kid)
        except NotImplementedError:
            continue
        y2 = kernels.gemm_forward(x, qw, sc, qz, kernel_id=kid)
        torch.cuda.synchronize()
        err = (y.float() - ref).abs().max().item() / scale
        same = bool((y == y2).all())
        eqw = bool((y == wide).all()) if (s == 1 and mb >= 4) else None
        nanc = int(torch.isnan(y.float()).sum())
        okk = err <= 2e-3 and same and nanc == 0 and (eqw is not False)
        bad += not okk
        print(f"{spec} S={s}: rel err {err:.2e} rerun-identical {same} bit-equal-to-wide {eqw} nan {nanc} {'ok' if okk else 'FAIL'}   [{plan}]", flush=True)
        if not okk and err > 2e-3:
            d = (y.float() - ref).abs()
            rows = (d.max(dim=1).values > 2e-3 * scale).nonzero().flatten()
            cols = (d.max(dim=0).values > 2e-3 * scale).nonzero().flatten()
            print("   bad rows", rows[:12].tolist(), "... count", len(rows), " bad cols", cols[:12].tolist(), "... count", len(cols), flush=True)
print("FAILURES", bad)
sys.exit(1 if bad else 0)
