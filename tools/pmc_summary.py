"""Summarise gpurun_out/pmc{1,2,3} counter CSVs for fk_play_kernel."""
import csv, glob, collections, sys
agg = {}
for d in ("pmc1", "pmc2", "pmc3"):
    fs = sorted(glob.glob(f"gpurun_out/{d}/runc/*_counter_collection.csv"))
    if not fs: continue
    rows = list(csv.DictReader(open(fs[-1])))
    per = collections.defaultdict(list)
    for r in rows:
        if "play" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        agg[k] = sum(v) / len(v)
games = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
rolls = games * 145.4
w = agg["SQ_WAVES"]
print(f"waves {w:.0f}")
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"):
    if k in agg: print(f"{k:22s} {agg[k]:.4g}  per ideal wave-iteration {agg[k] * 64 / rolls:8.1f}")
if "SQ_THREAD_CYCLES_VALU" in agg:
    print("lane utilisation of VALU instructions:", agg["SQ_THREAD_CYCLES_VALU"] / (agg["SQ_ACTIVE_INST_VALU"] * 64))
wc = agg.get("SQ_WAVE_CYCLES")
for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_SALU"):
    if k in agg and wc: print(f"{k:22s} {agg[k] / wc:6.3f} of wave cycles")
for k in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
    if k in agg: print(k, f"{agg[k]:.4g}")
