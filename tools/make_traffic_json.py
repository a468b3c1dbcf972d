"""HBM-traffic record of one PMC run of tools/pmc_cfg.sh, stamped with the kernel sources it was taken on.
usage: python tools/make_traffic_json.py <tag> <config> <algorithmic_bytes_per_launch> "<source command>" [round=3] [k]
(with k: profiles/r<round>_hbm_traffic_config<config>_k<k>.json, one stamp per player count of the sweep)
Writes profiles/r<round>_hbm_traffic_config<config>.json; bench.py attaches `hbm_bytes_corrected` as roofline.traffic only while
the sha256 of the kernel sources still matches."""
import collections, csv, glob, os, json, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import kernel_source_sha

tag, config, algo, source = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
rnd = int(sys.argv[5]) if len(sys.argv) > 5 else 3
k_suffix = f"_k{int(sys.argv[6])}" if len(sys.argv) > 6 else ""
agg = collections.defaultdict(dict)
def newest(paths):
    """one file per directory: the most recent run (gpurun merges every run's files into the same directory)"""
    best = {}
    for q in paths:
        d = os.path.dirname(q)
        if d not in best or os.path.getmtime(q) > os.path.getmtime(best[d]):
            best[d] = q
    return sorted(best.values())


for f in newest(glob.glob(str(ROOT / f"gpurun_out/{tag}_pmc[456]/**/*_counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = next((k for k in ("fk_play_hc_kernel", "fk_play_kernel", "fk_seed_kernel", "fk_perm_parallel", "fk_perm_draw", "fk_perm_kernel", "fk_tally_reduce")
                     if k in r["Kernel_Name"]), None)
        if name:
            per[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, counter), v in per.items():
        agg[name][counter] = sum(v) / len(v)
kernels = {}
for name, c in agg.items():
    rec = {f"{k}_KiB" if k.endswith("SIZE") else k: v for k, v in c.items()}
    if "FETCH_SIZE" in c:
        rec["hbm_bytes_raw"] = (c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024
        rec["hbm_bytes_corrected"] = (2 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024
    kernels[name] = rec
game_kernel = max((n for n in ("fk_play_hc_kernel", "fk_play_kernel") if n in kernels), key=lambda n: kernels[n].get("hbm_bytes_corrected", 0.0),
                  default="fk_play_kernel")  # the game kernel the launch plan chose for this configuration
kernels.setdefault(game_kernel, {})["algorithmic_bytes"] = algo
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
out = {"source": source, "round": rnd, "config": config, "commit": head, "kernel_source_sha256": kernel_source_sha(),
       "units": "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; hbm_bytes_corrected = 2 x FETCH_SIZE + WRITE_SIZE "
                "(the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md applied to the whole read side: an upper bound for this "
                "access mix); per launch, mean over the launches of the run",
       "kernels": kernels}
path = ROOT / "profiles" / f"r{rnd:02d}_hbm_traffic_config{config}{k_suffix}.json"
path.write_text(json.dumps(out, indent=1) + "\n")
print(path, game_kernel, json.dumps(kernels.get(game_kernel), indent=0)[:400])
