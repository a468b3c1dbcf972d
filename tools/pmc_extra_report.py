"""Mean of every counter of tools/pmc_extra.sh over the game kernel's dispatches.  usage: python tools/pmc_extra_report.py <tag> [kernel-substring]"""
import collections, csv, glob, os, sys
tag = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "fk_play"
best = {}
for q in glob.glob(f"gpurun_out/{tag}_x*/**/*_counter_collection.csv", recursive=True):
    d = os.path.dirname(q)
    if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
        best[d] = q
for f in sorted(best.values()):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, v in per.items():
        print(f"{tag:10s} {name:32s} {sum(v) / len(v):14.6g}  ({len(v)} dispatches)")
