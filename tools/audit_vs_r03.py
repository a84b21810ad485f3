"""Report of scripts/archive/r04_gpu_audit_vs_r03.sh: this tree's planner pick against r03's library, per shape (two alternating rounds each)."""
import json, sys, collections, glob, os, math
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04")
t = collections.defaultdict(lambda: collections.defaultdict(list))
plan = {}
for f in glob.glob(os.path.join(d, "audit_*_*.jsonl")) + glob.glob(os.path.join(d, "r04_audit_r0*_lib.jsonl")):   # (gpurun_out/r04 or profiles)
    which = "r03" if "audit_r03" in f else "new"
    for l in open(f):
        if l.startswith("{"):
            r = json.loads(l)
            t[r["shape"]][which].append(r["kernel_us"])
            plan[(r["shape"], which)] = " ".join(r["plan"].split()[:3])
key = lambda s: (int(s.split("x")[1]), int(s.split("x")[2]), int(s.split("x")[0]))
logs, worst = [], (0, "")
for sh in sorted(t, key=key):
    a, b = min(t[sh]["r03"]), min(t[sh]["new"])
    r = b / a
    logs.append(math.log(r))
    if r > worst[0]: worst = (r, sh)
    mark = "  <-- slower" if r > 1.03 else ("  faster" if r < 0.97 else "")
    print(f"{sh:>18s}  r03 {a:8.2f} us [{plan[(sh, 'r03')]:40s}]  r04 {b:8.2f} us [{plan[(sh, 'new')]:40s}]  {r:.3f}{mark}")
print(f"{len(logs)} shapes: geometric mean r04 / r03 = {math.exp(sum(logs) / len(logs)):.4f}; worst {worst[0]:.3f} at {worst[1]}")
