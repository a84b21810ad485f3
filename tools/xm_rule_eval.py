"""Evaluates mid-token planner rules against an audit file written by tools/xm_audit.py: for every (M, K, N) the time of the rule's pick relative to the
best of (the other families' pick, every XM configuration).   python tools/xm_rule_eval.py gpurun_out/r06/xm_audit.txt [--list]"""
import re, sys, math
CUS = 256
rows = []
for l in open(sys.argv[1]):
    m = re.match(r"\s*(\d+) x\s*(\d+) x\s*(\d+)\s+others\s+([\d.]+) us \[(.*?)\]\s+xm best.*all: (.*)", l)
    if not m:
        continue
    M, K, N, base = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
    opts = {}
    for tok in m.group(6).split():
        k, v = tok.split(":")
        pr, t32 = int(k[0]), k.endswith("t")
        mb = 1 if (M <= 32 or t32) else 2
        opts[(mb, pr)] = float(v)
    rows.append((M, K, N, base, m.group(5), opts))


def one_round(pairs, mt):
    """smallest number of channel pairs per workgroup (1..3) that covers the layer in one round of workgroups, or 0"""
    for pr in (1, 2, 3):
        if -(-pairs // pr) * mt <= CUS:
            return pr
    return 0


def rule(M, K, N):
    """-> (mb, pr) or None: the rule of make_plan (w4a16_gemm.hip), restated"""
    pairs, KT = N // 32, K // 128
    if KT < 32 or M <= 16 or M > 128:
        return None
    if M > 64:
        p1, p2 = one_round(pairs, -(-M // 32)), one_round(pairs, 2)
        if p1 and p1 <= 2 and (KT <= 64 or (KT <= 86 and M <= 80)):
            return (1, p1)
        if p1 == 3 and M <= 96 and KT <= 64:
            return (1, 3)
        if p2 and 10 * (-(-pairs // p2) * 2) >= 8 * CUS and p2 <= 2 and KT <= 64:
            return (2, p2)
        return None
    if M <= 32:
        pr = one_round(pairs, 1)
        return (1, pr) if pr and KT <= 64 else None
    if 2 * pairs <= CUS:                                  # two 32-token tiles x one pair: every CU busy beats the halved dequantisation
        return (1, 1) if KT <= 64 or (KT <= 86 and M <= 48) else None
    if KT > 64:
        return None
    p1, p2 = one_round(pairs, 2), one_round(pairs, 1)
    if p1 and p1 <= 2:
        pick = (1, p1)
    elif p2:
        pick = (2, p2)
    else:
        return None
    wgs = -(-pairs // pick[1]) * (2 if pick[0] == 1 else 1)
    if pick[0] == 1 and 10 * wgs < 7 * CUS and M < 56:   # (two 32-token tiles on a layer that leaves CUs idle: the fragment kernels stay ahead up to 55 tokens)
        return None
    if pick[0] == 2 and KT > 32 and M > 56 and 10 * wgs < 7 * CUS:   # (64-token tiles, long K, > 30 % of the CUs idle: behind the 64 x 128 four-wave tile from 57 tokens -- 64 x 8192 x 10240 on three boxes)
        return None
    return pick


tot, n, worst = 0.0, 0, []
for M, K, N, base, plan, opts in rows:
    pick = rule(M, K, N)
    t = opts.get(pick, base) if pick else base
    ideal = min([base] + list(opts.values()))
    tot += math.log(t / ideal); n += 1
    worst.append((t / ideal, M, K, N, pick, t, base, min(opts, key=opts.get), min(opts.values())))
    if "--list" in sys.argv:
        print(f"{M:3d} x {K:5d} x {N:5d}: others {base:6.2f}  pick {pick} {t:6.2f}  best xm {min(opts, key=opts.get)} {min(opts.values()):6.2f}   vs others {t / base:5.3f}  vs ideal {t / ideal:5.3f}")
worst.sort(reverse=True)
print(f"{n} shapes: geomean pick / ideal {math.exp(tot / n):.4f}; worst:")
for w in worst[:12]:
    print("   ratio %.3f  %d x %d x %d pick %s %.2f us (others %.2f, best xm %s %.2f)" % w)
