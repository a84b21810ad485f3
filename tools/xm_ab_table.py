"""Tabulates an A/B file written by scripts/r06/gpu_xm_ab.sh: one row per (shape, configuration), one column per variant, the smaller of the two rounds.
    python tools/xm_ab_table.py gpurun_out/r06/xm_ab_<tag>.txt [span|dispatch]"""
import collections, re, sys
which = sys.argv[2] if len(sys.argv) > 2 else "span"
rows, cur = collections.OrderedDict(), None
for l in open(sys.argv[1]):
    if l.startswith("=="):
        cur = l.split()[1]
        continue
    m = re.match(r"\s+(\S+) xm (pr=\d(?: t32| big)?)\s+span\s+([\d.]+|nan) us\s+dispatch\s+([\d.]+|nan)", l)
    if m:
        rows.setdefault((m.group(1), m.group(2)), collections.OrderedDict()).setdefault(cur, []).append(float(m.group(3 if which == "span" else 4)))
vs = None
for k, d in rows.items():
    if vs is None:
        vs = list(d)
        print("%-30s" % (which + " us (min of rounds)") + "".join("%10s" % v for v in vs))
    print("%-30s" % " ".join(k) + "".join("%10.2f" % min(d.get(v, [float("nan")])) for v in vs))
