"""Evidence for the DVFS statement of DESIGN.md 5.6: the same launch -- identical instructions and addresses -- with the activations
N(0, 0.5), all ones, all zeros, while a thread samples the GPU's shader clock and socket power from sysfs (hwmon) at ~20 Hz;
rocm-smi as a fallback at whatever rate it answers.   python tools/power_log.py [MxKxN] [seconds per fill]"""
import glob, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quick_amd import packing, gemm_forward, kernels

spec = sys.argv[1] if len(sys.argv) > 1 else "4096x8192x8192"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
M, K, N = (int(v) for v in spec.split("x"))
dev = torch.device("cuda:0")


def sysfs_sources():
    out = {}
    for hw in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name, key in (("power1_average", "power_uW"), ("power1_input", "power_uW"), ("freq1_input", "sclk_Hz")):
            p = os.path.join(hw, name)
            if os.path.exists(p) and key not in out:
                out[key] = p
    return out


SRC = sysfs_sources()
samples, stop = [], threading.Event()


def sample_loop():
    while not stop.is_set():
        t = time.perf_counter()
        rec = {"t": t}
        if SRC:
            for key, p in SRC.items():
                try:
                    rec[key] = float(open(p).read().strip())
                except Exception:
                    pass
        else:
            try:
                j = json.loads(subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5).stdout)
                card = next(iter(j.values()))
                for k, v in card.items():
                    if "sclk" in k.lower() and "(" in str(v):
                        rec["sclk_Hz"] = float(str(v).split("(")[1].split("M")[0]) * 1e6
                    if "power" in k.lower():
                        rec["power_uW"] = float(v) * 1e6
            except Exception:
                pass
        samples.append(rec)
        time.sleep(0.05)


qw, sc, qz = packing.random_mi355x(K, N, 128, dev)
y = torch.empty(M, N, device=dev, dtype=torch.float16)
print(f"{spec}: {kernels.plan_describe(M, K, N, 128)}\nsources: {SRC or 'rocm-smi'}")
th = threading.Thread(target=sample_loop, daemon=True)
th.start()
for fill in ("randn", "ones", "zeros", "randn"):
    x = (torch.randn(M, K, device=dev) * 0.5).half()
    if fill == "ones":
        x.fill_(1.0)
    elif fill == "zeros":
        x.zero_()
    for _ in range(5):
        gemm_forward(x, qw, sc, qz, out=y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            gemm_forward(x, qw, sc, qz, out=y)
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    us = e0.elapsed_time(e1) * 1e3 / n
    mine = [s for s in samples if t0 + 0.5 <= s["t"] <= t1]
    clk = [s["sclk_Hz"] / 1e9 for s in mine if "sclk_Hz" in s]
    pw = [s["power_uW"] / 1e6 for s in mine if "power_uW" in s]
    print(f"x = {fill:5s}: {us:8.1f} us per launch = {2.0 * M * N * K / us / 1e6:7.0f} TFLOP/s over {n} launches; "
          f"{len(mine)} samples: sclk mean {np.mean(clk) if clk else float('nan'):.3f} GHz (min {min(clk) if clk else float('nan'):.3f}, max {max(clk) if clk else float('nan'):.3f}), "
          f"socket power mean {np.mean(pw) if pw else float('nan'):.0f} W (max {max(pw) if pw else float('nan'):.0f})")
stop.set()
