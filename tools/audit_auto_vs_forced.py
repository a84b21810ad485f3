"""Offline: the planner's pick (variants auto, auto2: the minimum of the two passes) against the best FORCED launch of every family per shape,
from a tools/wide_probe.py --out file (scripts/r05/gpu_planner_audit.sh).  A forced variant that resolves to the same plan text as AUTO
counts as AUTO's time too (same launch, another sample).
    python tools/audit_auto_vs_forced.py gpurun_out/r05/planner_audit.jsonl [more.jsonl ...] > profiles/r05_planner_audit.txt
With several files (sessions / boxes) a shape's gap is the MEDIAN over the files of auto / best."""
import collections, json, re, sys
import numpy as np

def load(path):
    by = collections.OrderedDict()
    for l in open(path):
        r = json.loads(l)
        by.setdefault(r["shape"], {})[r["variant"]] = r
    return by

def short(plan):
    m = re.match(r"(\w+) (?:tokens=(\d+) channels=(\d+)|ntw=(\d+) waves=(\d+))", plan)
    s = re.search(r"(slices|ksplit)=(\d+)", plan)
    tail = f"{s.group(1)}={s.group(2)}" if s else "slices=1"      # (the mid-token kernels: all of K in one workgroup)
    if m.group(2):
        return f"{m.group(1)} tokens={m.group(2)} channels={m.group(3):<11s} {tail}"
    return f"{m.group(1)} ntw={m.group(4)} waves={m.group(5):<15s} {tail}"

files = [load(p) for p in sys.argv[1:]]
rows = []
for shape in files[0]:
    gaps, line = [], None
    for by in files:
        v = by.get(shape)
        if not v or "auto" not in v:
            continue
        plan = v["auto"]["plan"]
        auto = min(x["kernel_us"] for n, x in v.items() if x["plan"] == plan)
        forced = {n: x["kernel_us"] for n, x in v.items() if not n.startswith("auto") and x.get("kernel_us")}
        bn = min(forced, key=forced.get)
        best = min(forced[bn], auto)
        gaps.append(auto / best)
        if line is None:
            line = (plan, auto, bn, forced[bn])
    g = float(np.median(gaps))
    rows.append((shape, line, g, gaps))
gs = np.array([r[2] for r in rows])
print(f"# planner audit: AUTO (min over its samples) against forced launches of every family, {len(rows)} shapes, {len(files)} session(s)")
print(f"# geometric mean auto / best = {np.exp(np.mean(np.log(gs))):.4f}; worst {gs.max():.3f}; shapes with a gap > 3 %: {(gs > 1.03).sum()}, > 5 %: {(gs > 1.05).sum()}")
for shape, (plan, auto, bn, bt), g, gaps in rows:
    extra = ("  [" + " ".join(f"{x:.3f}" for x in gaps) + "]") if len(gaps) > 1 else ""
    print(f"{shape:>18s} auto {auto:8.2f} us [{short(plan):44s}] best forced {bn:8s} {bt:8.2f}  auto / best {g:.3f}{'  <-- ' if g > 1.03 else ''}{extra}")
