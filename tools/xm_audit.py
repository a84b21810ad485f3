"""Planner audit of the mid-token kernels: every (tile, channel pairs) selection of QUICK_KERNEL_XM against the other families' pick
(QUICK_AMD_XM=0 in the environment) on the decode layer shapes, dispatch clock, HBM-cold weight sets.
    QUICK_AMD_XM=0 python tools/xm_audit.py [--M 17,24,32,33,48,64] [KxN ...] > audit.txt"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
XM = 7
args = sys.argv[1:]
Ms = [17, 24, 32, 33, 40, 48, 56, 64]
if args and args[0] == "--M":
    Ms = [int(v) for v in args[1].split(",")]
    args = args[2:]
LAYERS = args or ["4096x4096", "4096x12288", "4096x22016", "11008x4096", "4096x6144", "4096x28672", "14336x4096", "8192x8192", "8192x10240", "8192x57344", "28672x8192",
                  "5120x5120", "5120x15360", "5120x27648", "13824x5120"]


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
        base = timed(M, K, N, 0, sets, x, y, ws)
        row = {}
        for pr in (1, 2, 3):
            for t32 in ((0, 1) if M > 32 else (0,)):
                row[(pr, t32)] = timed(M, K, N, XM | (pr << 4) | (t32 << 8), sets, x, y, ws)
        best = min(row, key=lambda k: row[k])
        print(f"{M:3d} x {K:5d} x {N:5d}  others {base:7.2f} us [{kernels.plan_describe(M, K, N, G).split(' grid')[0]}]   xm best pr={best[0]} {'2x32' if best[1] else ('64' if M > 32 else '32')} {row[best]:7.2f} us  ratio {row[best] / base:5.3f}   all: "
              + "  ".join(f"{k[0]}{'t' if k[1] else ''}:{v:.2f}" for k, v in row.items()), flush=True)
    del sets
