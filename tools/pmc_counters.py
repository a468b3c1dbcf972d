"""Average the counters of every pass directory gpurun_out/<tag>* for kernels matching a substring.
usage: python tools/pmc_counters.py <tag-prefix> [kernel-substring]"""
import collections, csv, glob, os, sys
tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "fk_play"
best = {}
for q in glob.glob(f"gpurun_out/{tag}*/**/*_counter_collection.csv", recursive=True):
    d = os.path.dirname(q)
    if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
        best[d] = q
for d, f in sorted(best.items()):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(per.items()):
        print(f"{d.split('/')[1]:14s} {k:34s} {sum(v) / len(v):14.6g}  ({len(v)} dispatches)")
