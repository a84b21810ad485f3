"""Diagnostic (not product): element-wise error of the tiled kernel at M=512 and the y(2x) == 2y(x) property."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, quick_amd
dev = torch.device("cuda:0")
K = N = 4096; G = 128; M = 512
x, iw, s, z = oracle.make_synthetic(M, K, N, G, seed=0)
x[np.abs(x) < 2.0 ** -13] = 2.0 ** -13
wdeq = oracle.dequantize(iw, s, z, G)
ref32 = x.astype(np.float32) @ wdeq.astype(np.float32)
packed = [torch.from_numpy(a).to(dev) for a in oracle.pack_mi355x(iw, s, z)]
xd = torch.from_numpy(x).to(dev)
for name, kid in (("tiled", 2), ("skinny", 1)):
    y1 = quick_amd.gemm_forward(xd, *packed, kernel_id=kid)
    y2 = quick_amd.gemm_forward(xd * 2, *packed, kernel_id=kid)
    d = (y1.float().cpu().numpy() - ref32)
    ulp = np.abs(d) / (np.abs(ref32) * 2.0 ** -11 + 1e-6)
    print(name, "max abs diff", np.abs(d).max(), "at", np.unravel_index(np.abs(d).argmax(), d.shape), "max err in fp16 ulps of ref", ulp.max(),
          "frac > 1 ulp", (ulp > 1.0).mean())
    mism = (y2 != y1 * 2)
    print(name, "2x mismatches", int(mism.sum().item()), "of", mism.numel())
    if mism.any():
        idx = mism.nonzero()[:10].cpu().numpy()
        for r, c in idx:
            print("   ", r, c, float(y1[r, c]), float(y2[r, c]), ref32[r, c])
        rows = mism.any(1).nonzero().flatten().cpu().numpy(); cols = mism.any(0).nonzero().flatten().cpu().numpy()
        print("    rows with mismatch:", len(rows), rows[:20], " cols:", len(cols), cols[:20])
