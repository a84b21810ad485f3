"""Offline: compare the planner's choice with the best variant per shape in a tools/wide_probe.py --out file.
The planner's own row ("auto") is measured first for every shape and reads up to ~10 % high (clocks ramping after the
allocation pause), so where a variant has the same plan text, that variant's time stands in for it."""
import collections, json, sys
import numpy as np
rows = [json.loads(l) for l in open(sys.argv[1])]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
by = collections.defaultdict(dict)
for r in rows:
    by[tuple(int(v) for v in r["shape"].split("x"))][r["variant"]] = r
gaps = []
for k, v in sorted(by.items(), key=lambda kv: (kv[0][1], kv[0][2], kv[0][0])):
    a = v["auto"]
    v.pop("warm", None)
    same = [x["kernel_us"] for n, x in v.items() if n != "auto" and x["plan"] == a["plan"]]
    at = min(same) if same else a["kernel_us"]
    best = min(((n, x["kernel_us"]) for n, x in v.items() if n != "auto"), key=lambda t: t[1])
    if at < best[1]:
        best = ("auto", at)
    gap = at / best[1] - 1
    gaps.append(gap)
    print(f"{k[0]:5d}x{k[1]:5d}x{k[2]:5d} auto {at:8.1f}{'=' if same else '?'} ({a['plan'][:44]}) best {best[0]:8s} {best[1]:8.1f} gap {gap * 100:4.1f}%  "
          f"{2 * k[0] * k[1] * k[2] / best[1] / 1e6:6.0f} TF{'  <<<' if gap > thr else ''}")
print(f"mean gap {np.mean(gaps) * 100:.2f} %  max {np.max(gaps) * 100:.1f} %  shapes {len(gaps)}")
