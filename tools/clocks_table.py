"""The clocks of the bench rows side by side: rocprofv3 kernel-trace average of the same bench command, the event-pair dispatch clock, the back-to-back step,
the in-kernel span, the empty-kernel floor.   python tools/clocks_table.py <bench.json> <bench_kernel_stats.csv> > profiles/r06_clocks.txt"""
import csv, json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
stats = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}


def trace(substrs):
    for name, r in stats.items():
        if all(s in name for s in substrs):
            return float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["MinNs"]) / 1e3
    return None, 0, None


SYM = {1: ("w4a16_lean_kernelILi8ELi4ELi1ELi0ELi0ELb0ELi1ELi1E",), 8: ("w4a16_lean_kernelILi8ELi4ELi1ELi0ELi0ELb0ELi16ELi1E",), 64: ("w4a16_xm_kernelILi1ELi1ELi0E",),
       512: ("w4a16_xw_kernelILi4ELi1ELi2ELi0E",)}
print("# the bench rows on every clock (us).  trace = rocprofv3 --kernel-trace --stats average over all launches of that kernel in the bench command (HBM-cold sweeps, warm-ups and the")
print("# decode-layer / clock-warm-up launches of the same kernel included: min in brackets); pairs = hipEvent pairs around single launches (bench `kernel_us`); step = back-to-back")
print("# graph-replayed step; span = per-wave s_memrealtime stamps, first wave in to last wave out (a span-stamped build of the same kernel); floor = an empty kernel on the pairs clock.")
print(f"# empty-kernel floor: {d.get('empty_kernel_us') or d['roofline'].get('empty_kernel_us')}")
print(f"{'row':>22s} {'plan':44s} {'trace':>8s} {'(min)':>8s} {'pairs':>8s} {'step':>8s} {'span':>8s}   frac(pairs)  frac(span)  traffic / algorithmic")
for r in d["sweep"]:
    ro = r["roofline"]
    t, calls, tmin = trace(SYM.get(r["M"], ("nothing",)))
    tr = ro.get("traffic")
    print(f"{'M = %d (4096 x 4096)' % r['M']:>22s} {ro['plan'][:44]:44s} {t if t else float('nan'):8.2f} {tmin if tmin else float('nan'):8.2f} {ro['kernel_us']:8.2f} {r['ms_per_step'] * 1e3:8.2f} "
          f"{ro.get('kernel_us_inkernel') or float('nan'):8.2f}   {ro['frac']:10.3f} {ro.get('frac_inkernel') or float('nan'):11.3f}  "
          + (f"{tr / 1e6:.2f} / {ro['algorithmic_bytes'] / 1e6:.2f} MB = {tr / ro['algorithmic_bytes']:.2f} x" if tr else "-"))
for r in d.get("decode_layers", []):
    ro = r["roofline"]
    tr = ro.get("traffic")
    print(f"{'%d x %d x %d' % (r['M'], r['K'], r['N']):>22s} {ro.get('plan', '')[:44]:44s} {'':8s} {'':8s} {r['kernel_us']:8.2f} {'':8s} {ro.get('kernel_us_inkernel') or float('nan'):8.2f}   {ro['frac']:10.3f} "
          f"{ro.get('frac_inkernel') or float('nan'):11.3f}  " + (f"{tr / 1e6:.2f} / {ro['algorithmic_bytes'] / 1e6:.2f} MB = {tr / ro['algorithmic_bytes']:.2f} x" if tr else "-"))
m = d.get("small_m_model")
if m:
    print(f"# small_m_model ({m['what']}): dispatch clock {m.get('dispatch_clock')}; in-kernel span {m.get('inkernel_span')}")
