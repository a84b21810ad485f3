"""Summarise rocprofv3 --pmc csv passes: mean counter value per dispatch for each W4A16 kernel."""
import csv, glob, os, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "quick_amd" not in k:
            continue
        k = k.split("(")[0].replace("void quick_amd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(d, "*kernel_trace.csv"))):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "quick_amd" in k:
            dur[k.split("(")[0].replace("void quick_amd::", "")].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k in agg:
    print(f"== {k}: {len(dur[k])} dispatches traced, mean duration under PMC {sum(dur[k]) / max(1, len(dur[k])):.0f} ns")
    for c, v in sorted(agg[k].items()):
        skip = v[3:] if len(v) > 6 else v
        print(f"   {c:36s} {sum(skip) / len(skip):16.1f}   (n={len(v)})")
