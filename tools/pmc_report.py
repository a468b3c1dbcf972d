"""Summarise the passes of tools/pmc_cfg.sh for the game kernel.
usage: python tools/pmc_report.py <tag> <rolls_per_launch> [kernel-substring]
Counters are averaged over the dispatches of the kernel (one launch each)."""
import collections, csv, glob, os, sys

tag, rolls = sys.argv[1], float(sys.argv[2])
sub = sys.argv[3] if len(sys.argv) > 3 else "fk_play"
agg: dict[str, float] = {}
def newest(paths):
    """one file per directory: the most recent run (gpurun merges every run's files into the same directory)"""
    best = {}
    for q in paths:
        d = os.path.dirname(q)
        if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
            best[d] = q
    return sorted(best.values())


for f in newest(glob.glob(f"gpurun_out/{tag}_pmc*/**/*_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        agg[k] = sum(v) / len(v)
for f in newest(glob.glob(f"gpurun_out/{tag}_stats/**/*_kernel_stats.csv", recursive=True)):
    print("kernel stats:", f)
    for r in csv.DictReader(open(f)):
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:9.3f} ms min {float(r['MinNs'])/1e6:9.3f} pct {r['Percentage']}")
print(f"rolls per launch {rolls:.4g}; per wave-roll = counter * 64 / rolls")
if "SQ_WAVES" in agg:
    print(f"waves {agg['SQ_WAVES']:.0f}")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
    if k in agg:
        print(f"{k:22s} {agg[k]:.4g}  per wave-roll {agg[k] * 64 / rolls:8.1f}")
if "SQ_THREAD_CYCLES_VALU" in agg and "SQ_ACTIVE_INST_VALU" in agg:
    print(f"VALU lane utilisation {agg['SQ_THREAD_CYCLES_VALU'] / (agg['SQ_ACTIVE_INST_VALU'] * 64):.4f}")
wc = agg.get("SQ_WAVE_CYCLES")
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS",
          "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU"):
    if k in agg and wc:
        print(f"{k:22s} {agg[k] / wc:6.3f} of wave cycles")
for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum",
          "TCC_MISS_sum"):
    if k in agg:
        print(f"{k:22s} {agg[k]:.5g}")
if "FETCH_SIZE" in agg:
    print(f"HBM read bytes raw {agg['FETCH_SIZE'] * 1024:.4g} (x2-corrected upper bound {agg['FETCH_SIZE'] * 2048:.4g}); "
          f"written {agg.get('WRITE_SIZE', 0) * 1024:.4g}")
