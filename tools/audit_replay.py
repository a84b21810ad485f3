"""Replay a planner audit on the CPU: for every shape of a `tools/wide_probe.py --jsonl` audit file, ask the library
what it would launch NOW (quick_w4a16_plan_describe is host-only) and look that launch up among the variants the
audit measured on the GPU.  Prints the mean / worst gap to the best measured variant and the shapes whose plan was
not among the measured ones.  Usage: python tools/audit_replay.py profiles/archive/r03_xk_audit.jsonl [--top 15]"""
import argparse
import collections
import json
import re

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quick_amd import kernels  # noqa: E402


def key(plan):
    return re.sub(r"\s*workspace=\d+", "", plan).strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("audit")
    ap.add_argument("--top", type=int, default=15)
    ap.add_argument("--group", type=int, default=128)
    a = ap.parse_args()
    by = collections.defaultdict(dict)
    for line in open(a.audit):
        r = json.loads(line)
        if r["variant"] == "warm":
            continue
        by[r["shape"]].setdefault(key(r["plan"]), r["kernel_us"])
        by[r["shape"]]["@auto"] = r["kernel_us"] if r["variant"] == "auto" else by[r["shape"]].get("@auto")
    gaps, old, missing = [], [], []
    for shape, d in by.items():
        M, K, N = map(int, shape.split("x"))
        now = key(kernels.plan_describe(M, K, N, a.group))
        best = min(v for k, v in d.items() if k != "@auto" and v)
        old.append(d["@auto"] / best - 1)
        if now in d:
            gaps.append((d[now] / best - 1, shape, d[now], best, now[:70]))
        else:
            missing.append((shape, now[:90]))
    gaps.sort(reverse=True)
    print(f"{len(by)} shapes; audited picks: mean gap {100 * sum(old) / len(old):.2f} % worst {100 * max(old):.1f} %")
    print(f"plans of this build found among the measured variants: {len(gaps)}; mean gap {100 * sum(g[0] for g in gaps) / len(gaps):.2f} %"
          f" worst {100 * gaps[0][0]:.1f} %")
    for g in gaps[: a.top]:
        print(f"  {100 * g[0]:5.1f} %  {g[1]:<18} {g[2]:7.1f} us (best {g[3]:.1f})  {g[4]}")
    print(f"not measured in the audit: {len(missing)}")
    for m in missing[: a.top]:
        print("  ", *m)


if __name__ == "__main__":
    main()
