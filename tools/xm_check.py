"""Diagnostic: the mid-token kernels (QUICK_KERNEL_XM = 7) against the oracle, and timed against the other families on HBM-cold weight sets
(dispatch clock and in-kernel span).
    python tools/xm_check.py [--no-check] [--no-time] [MxKxN ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
XM = 7
args = sys.argv[1:]
check = "--no-check" not in args
timeit = "--no-time" not in args
args = [a for a in args if not a.startswith("--")]
specs = args or ["64x4096x4096", "33x4096x4096", "17x1024x256", "64x1024x352", "50x1536x4096", "64x4096x12288", "64x4096x22016", "64x11008x4096", "32x4096x4096", "48x4096x14336"]


def arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def timed(M, K, N, kid, sets, x, y, ws):
    n = len(sets)
    qa, sa, za = arr([s[0] for s in sets]), arr([s[1] for s in sets]), arr([s[2] for s in sets])
    it = 48
    sp = (ctypes.c_float * it)()
    rc = lib.quick_w4a16_gemm_span(x.data_ptr(), qa, sa, za, n, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, kid, 0, it, sp, None)
    span = float(np.median(np.asarray(sp[:])[8:])) if rc == 0 else float("nan")
    it2 = 120
    us = (ctypes.c_float * it2)()
    rc = lib.quick_w4a16_gemm_profile(x.data_ptr(), qa, sa, za, n, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, kid, 0, it2, us, None)
    disp = float(np.median(np.asarray(us[:])[20:])) if rc == 0 else float("nan")
    return span, disp


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


bad = 0
for spec in specs:
    M, K, N = (int(v) for v in spec.split("x"))
    if check and K * N <= 4096 * 4096:
        x, iw, s, z = oracle.make_synthetic(M, K, N, G, seed=M + K + N)
        want = oracle.w4a16_forward(x, iw, s, z, G).astype(np.float32)
        packed = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in oracle.pack_mi355x(iw, s, z))
        xd = torch.from_numpy(x).to(dev)
        for pr in (1, 2, 3):
            for big in (0, 1, 2):
                kid = XM | (pr << 4) | (big << 8)
                try:
                    plan = kernels.plan_describe(M, K, N, G, kid)
                    y = kernels.gemm_forward(xd, *packed, kernel_id=kid)
                    y2 = kernels.gemm_forward(xd, *packed, kernel_id=kid)
                    torch.cuda.synchronize()
                except Exception as e:
                    print(f"{spec} pr={pr} big={big}: {type(e).__name__}: {e}")
                    bad += 1
                    continue
                e = rel(y.float().cpu().numpy(), want)
                same = bool(torch.equal(y, y2))
                flag = "" if (e <= 2e-3 and same) else "   <<<<<< WRONG"
                bad += bool(flag)
                print(f"{spec} pr={pr} big={big} [{plan}]: rel err {e:.2e} repeat-equal {same}{flag}", flush=True)
                if flag:
                    d = np.abs(y.float().cpu().numpy() - want)
                    rows = np.where(d.max(1) > 2e-3 * np.abs(want).max())[0]
                    cols = np.where(d.max(0) > 2e-3 * np.abs(want).max())[0]
                    print("   bad rows", rows[:40], "n", len(rows), " bad cols", cols[:64], "n", len(cols))
    if not timeit:
        continue
    nsets = max(2, min(40, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    ws = torch.zeros(48 << 20, dtype=torch.uint8, device=dev)
    algo = K * N / 2 + (K // G) * N * 2.5 + 2 * M * K + 2 * M * N
    os.environ["QUICK_AMD_XM"] = "0"
    timed(M, K, N, 0, sets, x, y, ws)
    base = () if "--only-xm" in sys.argv else (("auto", 0), ("skinny4", 1 | (4 << 4)))
    for name, kid in base + (("xm pr=1", XM | (1 << 4)), ("xm pr=2", XM | (2 << 4)), ("xm pr=3", XM | (3 << 4)), ("xm pr=1 t32", XM | (1 << 4) | (1 << 8)), ("xm pr=2 t32", XM | (2 << 4) | (1 << 8)), ("xm pr=3 t32", XM | (3 << 4) | (1 << 8)), ("xm pr=1 big", XM | (1 << 4) | (2 << 8)),
                      ("xm pr=2 big", XM | (2 << 4) | (2 << 8)), ("xm pr=3 big", XM | (3 << 4) | (2 << 8))):
        try:
            plan = kernels.plan_describe(M, K, N, G, kid)
        except Exception as e:
            print(f"   {name}: {e}")
            continue
        if (M > 32 and "big" in name) or (M <= 32 and "t32" in name):
            continue
        sp, dp = timed(M, K, N, kid, sets, x, y, ws)
        print(f"   {spec} {name:12s} span {sp:7.2f} us  dispatch {dp:7.2f} us  ({algo / dp / 8e6 * 100:5.1f}% of 8 TB/s, {2.0 * M * N * K / dp / 2.5e9 * 100:5.1f}% of 2.5 PF)  [{plan}]", flush=True)
print("WRONG RESULTS:" if bad else "all results right", bad)
