"""Report + fit of scripts/archive/r04_gpu_xw_sweep.sh (profiles/r04_xw_sweep.jsonl): per shape the planner's r03 pick against six forced four-wave
variants, the launch-time model of the four-wave kernels fitted to those rows (the coefficients in make_plan, w4a16_gemm.hip), and the
policy "four-wave kernel with the smallest estimate unless r03 picked the 256 x 256 tile" replayed on the measurements.
    python tools/xw_sweep_report.py [file]"""
import collections, json, math, os, re, sys
import numpy as np
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_xw_sweep.jsonl")
rows = [json.loads(l) for l in open(path) if l.strip().startswith("{")]

def geom(r):
    p = r["plan"]
    tok, ch = int(re.search(r"tokens=(\d+)", p).group(1)), int(re.search(r"channels=(\d+)", p).group(1))
    return tok // 32, ch // 128, int(re.search(r"grid=(\d+)", p).group(1)), int(re.search(r"slices=(\d+)", p).group(1))

def feats(K, grid, S):
    KT = K // 128
    st, f = -(-KT // S), grid / 256
    n = math.ceil(f)
    return [1, n, st * n, st * f, float(S > 1), S * (S > 1), f]

data, seen = collections.defaultdict(list), set()
for r in rows:
    if not r["plan"].startswith("xw"):
        continue
    M, K, N = map(int, r["shape"].split("x"))
    mb, pairs, grid, S = geom(r)
    if (M, K, N, mb, pairs, S) in seen:
        continue
    seen.add((M, K, N, mb, pairs, S))
    data[(mb, pairs)].append((feats(K, grid, S), r["kernel_us"]))
coef = {}
print("launch-time model  us = c + a n + stages (b_ceil n + b_frac f) + [S > 1] (s0 + s1 S) + d f,  f = workgroups / 256, n = ceil(f)")
for cfg, ds in sorted(data.items()):
    X, y = np.array([d[0] for d in ds]), np.array([d[1] for d in ds])
    c = np.linalg.lstsq(X / y[:, None], np.ones(len(y)), rcond=None)[0]
    err = X @ c / y - 1
    coef[cfg] = c
    print(f"  tile {cfg[0] * 32} x {cfg[1] * 128}: {len(ds)} rows, coefficients {np.round(c, 4).tolist()}, rms {np.sqrt((err ** 2).mean()) * 100:.1f} %, worst {np.abs(err).max() * 100:.1f} %")
by = collections.defaultdict(dict)
for r in rows:
    by[r["shape"]][r["variant"]] = r
key = lambda s: (int(s.split("x")[1]), int(s.split("x")[2]), int(s.split("x")[0]))
logs = []
for sh in sorted(by, key=key):
    d = by[sh]
    a = d["auto"]
    old = a["plan"].split()[0] + (re.search(r"tokens=(\d+)", a["plan"]).group(1) if "tokens=" in a["plan"] else "")
    est = {}
    for v, r in d.items():
        if v != "auto":
            mb, pairs, grid, S = geom(r)
            est[v] = float(np.array(feats(int(sh.split("x")[1]), grid, S)) @ coef[(mb, pairs)])
    pick = min(est, key=est.get)
    t = a["kernel_us"] if old == "wide256" else d[pick]["kernel_us"]
    logs.append(math.log(t / a["kernel_us"]))
    best = min(d, key=lambda v: d[v]["kernel_us"])
    print(f"{sh:>18s}  r03 pick {old:8s} {a['kernel_us']:8.2f} us | model's four-wave pick {pick:7s} est {est[pick]:7.1f} measured {d[pick]['kernel_us']:7.1f} | policy / r03 {t / a['kernel_us']:.2f} | best measured {best} {d[best]['kernel_us'] / a['kernel_us']:.2f}")
print(f"{len(logs)} shapes: geometric mean policy / r03 = {math.exp(sum(logs) / len(logs)):.3f}")
