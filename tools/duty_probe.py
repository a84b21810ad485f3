"""Is a mid-token launch bound by the chip's power management rather than by its schedule?  The in-kernel span of ONE launch (HBM-cold weight
sets) back to back, and with a one-thread spin kernel (torch.cuda._sleep: the GPU stays busy and clocked, at idle power) of growing length
between launches; shader clock and socket power sampled from sysfs meanwhile.   python tools/duty_probe.py MxKxN:kernel_id ..."""
import ctypes, glob, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import _lib, packing, kernels
lib = _lib.load()
dev = torch.device("cuda:0")
G = 128
SRC = {}
for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
    for name, key in (("power1_average", "power_uW"), ("power1_input", "power_uW"), ("freq1_input", "sclk_Hz")):
        p = os.path.join(hw, name)
        if os.path.exists(p) and key not in SRC:
            SRC[key] = p
samples, stop = [], threading.Event()


def sample_loop():
    while not stop.is_set():
        rec = {}
        for key, p in SRC.items():
            try:
                rec[key] = float(open(p).read().strip())
            except Exception:
                pass
        samples.append(rec)
        time.sleep(0.02)


def arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def mean_of(key, lo, hi):
    v = [s[key] for s in samples[lo:hi] if key in s]
    return float(np.mean(v)) if v else float("nan")


threading.Thread(target=sample_loop, daemon=True).start()
for spec in sys.argv[1:]:
    shape, _, kid = spec.partition(":")
    kid = int(kid or 0)
    M, K, N = (int(v) for v in shape.split("x"))
    nsets = max(2, min(40, int(400e6 / (K * N / 2)) + 1))
    sets = [packing.random_mi355x(K, N, G, dev) for _ in range(nsets)]
    y = torch.empty(M, N, dtype=torch.float16, device=dev)
    ws = torch.zeros(48 << 20, dtype=torch.uint8, device=dev)
    print(f"{shape} [{kernels.plan_describe(M, K, N, G, kid)}]", flush=True)
    for fill in ("randn", "zeros"):
        x = (torch.randn(M, K, device=dev) * 0.5).half() if fill == "randn" else torch.zeros(M, K, device=dev, dtype=torch.float16)
        for gap in (0, 20000, 100000, 400000, 1600000):   # spin cycles between launches (~0, 10, 50, 200, 800 us)
            spans = []
            i0 = len(samples)
            t_end = time.perf_counter() + 1.5
            k = 0
            while time.perf_counter() < t_end:
                if gap == 0:
                    it = 48
                    qa, sa, za = arr([s[0] for s in sets]), arr([s[1] for s in sets]), arr([s[2] for s in sets])
                    sp = (ctypes.c_float * it)()
                    rc = lib.quick_w4a16_gemm_span(x.data_ptr(), qa, sa, za, nsets, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, kid, 0, it, sp, None)
                    assert rc == 0, _lib.last_error()
                    spans += list(sp[8:])
                else:
                    st = sets[k % nsets]
                    k += 1
                    qa, sa, za = arr([st[0]]), arr([st[1]]), arr([st[2]])
                    sp = (ctypes.c_float * 1)()
                    torch.cuda._sleep(gap)
                    rc = lib.quick_w4a16_gemm_span(x.data_ptr(), qa, sa, za, 1, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, G, kid, 0, 1, sp, None)
                    assert rc == 0, _lib.last_error()
                    spans.append(sp[0])
            i1 = len(samples)
            print(f"   x {fill:5s} gap {gap:8d} cycles: span median {np.median(spans):6.2f} us  min {np.min(spans):6.2f}  n {len(spans):5d}   sclk {mean_of('sclk_Hz', i0, i1) / 1e6:7.0f} MHz  power {mean_of('power_uW', i0, i1) / 1e6:6.0f} W", flush=True)
stop.set()
