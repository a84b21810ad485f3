"""Audit of the AUTO rule that hands 5..16 tokens to the lean kernels (unmeasured for K > 4096): AUTO's pick against the skinny family's own
pick (kernel id SKINNY: the r01-r04 kernels) and the forced lean launch, dispatch clock, HBM-cold weight sets.
    python tools/lean_rule_audit.py [--M 5,6,8,12,16] [KxN ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
args = sys.argv[1:]
Ms = [5, 6, 8, 12, 16]
if args and args[0] == "--M":
    Ms = [int(v) for v in args[1].split(",")]
    args = args[2:]
LAYERS = args or ["4096x4096", "4096x8192", "5120x5120", "5120x8192", "8192x4096", "8192x8192", "8192x10240", "11008x4096", "11008x8192", "13824x5120", "14336x4096", "5120x15360", "4096x12288"]


def arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def timed(M, K, N, kid, sets, x, y, ws):
    n = len(sets)
    qa, sa, za = arr([s[0] for s in sets]), arr([s[1] for s in sets]), arr([s[2] for s in sets])
    it = 60
    us = (ctypes.c_float * it)()
    rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qa, sa, za, n, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, kid, 0, it, us, None)
    return float(np.median(np.asarray(us[:])[12:])) if rc == 0 else float("nan")


ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
for spec in LAYERS:
    K, N = (int(v) for v in spec.split("x"))
    nsets = max(2, min(24, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    for M in Ms:
        x = (torch.randn(M, K, device=dev) * 0.5).half()
        y = torch.empty(M, N, dtype=torch.float16, device=dev)
        timed(M, K, N, 0, sets, x, y, ws)
        t = {name: min(timed(M, K, N, kid, sets, x, y, ws), timed(M, K, N, kid, sets, x, y, ws)) for name, kid in (("auto", 0), ("skinny", 1), ("lean", 6))}
        plans = {name: kernels.plan_describe(M, K, N, G, kid).split(" grid")[0] if not np.isnan(t[name]) else "-" for name, kid in (("auto", 0), ("skinny", 1), ("lean", 6))}
        best = min(v for v in t.values() if not np.isnan(v))
        print(f"{M:3d} x {K:5d} x {N:5d}  auto {t['auto']:7.2f} [{plans['auto'][:44]:44s}]  skinny family {t['skinny']:7.2f} [{plans['skinny'][:50]:50s}]  lean {t['lean']:7.2f}   auto / best {t['auto'] / best:5.3f}", flush=True)
    del sets
