"""Summarise gpurun_out/<tag>_mem{1..5} (tools/pmc_mem.sh) for the game kernel: per-launch averages, and the counters that are
cycle counts as a fraction of what their block could count in the kernel's duration.
usage: python3 tools/pmc_mem_report.py <tag> [kernel-name substring, default play]"""
import collections, csv, glob, sys

tag = sys.argv[1]
which = sys.argv[2] if len(sys.argv) > 2 else "fk_play"
agg = {}
for i in range(1, 9):
    fs = sorted(glob.glob(f"gpurun_out/{tag}_mem{i}/**/*_counter_collection.csv", recursive=True))
    if not fs:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if which in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, v in per.items():
        agg[name] = sum(v) / len(v)
gui = agg.get("GRBM_GUI_ACTIVE", 0.0) / 8.0  # the counter is summed over the eight XCDs
print(f"{tag}: kernel duration {gui:.4g} cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs)")
CUS, XCD_CH = 256, 8 * 16
for name in sorted(agg):
    v = agg[name]
    extra = ""
    if gui and name in ("TA_TA_BUSY_sum", "TCP_GATE_EN1_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum",
                        "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"):
        extra = f"  = {v / (gui * CUS):.3f} of (CUs x duration)"
    if gui and name == "TCC_BUSY_sum":
        extra = f"  = {v / (gui * XCD_CH):.3f} of (L2 channels x duration)"
    if gui and name in ("TCC_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"):
        extra = f"  = {v / (gui * XCD_CH):.3f} per L2 channel and cycle"
    print(f"  {name:40s} {v:14.5g}{extra}")
if "TCP_TCC_READ_REQ_LATENCY_sum" in agg and agg.get("TCP_TCC_READ_REQ_sum"):
    print(f"  mean L1->L2 read latency {agg['TCP_TCC_READ_REQ_LATENCY_sum'] / agg['TCP_TCC_READ_REQ_sum']:.0f} cycles")
