"""Diagnostic: the lean small-M kernels (QUICK_KERNEL_LEAN = 6) against the oracle and against the planner's pick on the clock that can
see them -- the in-kernel span (first wave in -> last wave out, HBM-cold weight sets) -- plus the dispatch-clock duration.
    python tools/lean_check.py [--no-check] [MxKxN ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
LEAN = 6
args = sys.argv[1:]
check = "--no-check" not in args
planner_only = "--planner-only" in args
args = [a for a in args if not a.startswith("--")]
specs = args or ["1x4096x4096", "2x4096x4096", "4x4096x4096", "8x4096x4096", "16x4096x4096", "1x4096x12288", "1x4096x22016", "1x11008x4096"]


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


for spec in specs:
    M, K, N = (int(v) for v in spec.split("x"))
    nsets = max(2, min(40, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    ws = torch.zeros(32 << 20, dtype=torch.uint8, device=dev)
    algo = K * N / 2 + (K // G) * N * 2.5 + 2 * M * K + 2 * M * N
    timed(M, K, N, 0, sets, x, y, ws)   # (clocks, caches)
    base = timed(M, K, N, 0, sets, x, y, ws)
    print(f"{spec}: planner [{kernels.plan_describe(M, K, N, G)}] span {base[0]:.2f} us ({algo / base[0] / 8e6 * 100:.1f}% of 8 TB/s), dispatch {base[1]:.2f}")
    variants = () if planner_only else ((1, 8, 0), (1, 16, 0), (2, 8, 0), (2, 16, 0))
    if "--persist" in sys.argv:
        variants = ((1, 8, 0), (1, 8, 1), (1, 8, 2), (2, 8, 0), (2, 8, 1), (2, 8, 2))
    for ntw, waves, slots in variants:
        kid = LEAN | (ntw << 4) | ((waves // 4) << 8) | (slots << 22)
        try:
            plan = kernels.plan_describe(M, K, N, G, kid)
        except Exception as e:
            print(f"   ntw={ntw} waves={waves}: {e}")
            continue
        if "tiles_per_wave<=0" in plan:
            print(f"   ntw={ntw} waves={waves}: no build")
            continue
        err = ""
        if check:
            xs, iw, s, z = oracle.make_synthetic(M, K, min(N, 1024), G, seed=M + K)
            want = oracle.w4a16_forward(xs, iw, s, z, G)
            pk = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in oracle.pack_mi355x(iw, s, z)]
            got = kernels.gemm_forward(torch.from_numpy(xs).to(dev), *pk, kernel_id=kid).cpu().numpy().astype(np.float32)
            e = np.abs(got - want.astype(np.float32)).max() / np.abs(want.astype(np.float32)).max()
            err = f" rel_err {e:.2e}{'  <-- WRONG' if not e <= 2e-3 else ''}"
        t = timed(M, K, N, kid, sets, x, y, ws)
        print(f"   [{plan}] span {t[0]:.2f} us ({algo / t[0] / 8e6 * 100:.1f}%), dispatch {t[1]:.2f}{err}")
