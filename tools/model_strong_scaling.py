"""Model (no hardware needed): wall time of the strong-scaling bench configurations at N ranks from the measured single-GPU launches.

bench.py --config 4 / 6 split EVERY player count's shuffle range over the ranks (whole deterministic batches per rank).  The alternative
the round-5 review asked to weigh is k-major dealing: whole player counts per rank, no launch cut.  Inputs: profiles/r06_bench_config{4,6}.json
(per-k kernel time of one full-size launch, the measured drain tail = last workgroup's end - median workgroup's end, seed / permutation
kernels scale with the games and are inside the step time).  Model per rank and step:
    range split:  sum_k (t_k / N + tail_k + fixed)          fixed = 40 us of launch latency per engine call (6 kernels back to back)
    k-major:      longest-processing-time-first assignment of whole player counts to ranks, max over ranks of sum (t_k + tail_k + fixed)
Both end in one ncclReduce of the tally (1.07 MB at S = 5 160: ~0.1 ms over xGMI, modelled as 0.2 ms).
usage: python tools/model_strong_scaling.py [out.json]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
FIXED_MS, REDUCE_MS = 0.04, 0.2


def lpt(loads, n):
    bins = [0.0] * n
    for x in sorted(loads, reverse=True):
        bins[bins.index(min(bins))] += x
    return max(bins)


out = {}
for config in (4, 6):
    line = json.loads((ROOT / "profiles" / f"r06_bench_config{config}.json").read_text().strip().splitlines()[-1])
    per_k = line["roofline"]["per_k"]
    step_ms = line["ms_per_step"]
    kernel = {p["k"]: p["kernel_ms"] * (p.get("launches_per_step") or 1) for p in per_k}
    # everything of a step that is not a game kernel (seeding, permutations, tally post-passes, host) scales with the games like the kernels do
    other = step_ms - sum(kernel.values())
    tails = {p["k"]: float(p.get("launch_tail_ms") or 0.6) for p in per_k}
    rows = []
    for n in (1, 2, 4, 8):
        split = sum(t / n + tails[k] + FIXED_MS for k, t in kernel.items()) + other / n + (REDUCE_MS if n > 1 else 0.0)
        kmajor = lpt([t + tails[k] + FIXED_MS + other * t / sum(kernel.values()) for k, t in kernel.items()], n) + (REDUCE_MS if n > 1 else 0.0)
        ideal = step_ms / n
        rows.append({"ranks": n, "range_split_ms": split, "range_split_efficiency": ideal / split, "k_major_ms": kmajor, "k_major_efficiency": ideal / kmajor})
    out[f"config{config}"] = {"step_ms_1_gpu": step_ms, "kernel_ms_per_k": kernel, "launch_tail_ms_per_k": tails, "model": rows}
    print(f"config {config}: 1 GPU {step_ms:.0f} ms per step")
    for r in rows:
        print(f"   N = {r['ranks']}: range split {r['range_split_ms']:8.1f} ms (efficiency {r['range_split_efficiency']:.3f})   k-major {r['k_major_ms']:8.1f} ms ({r['k_major_efficiency']:.3f})")
out["conclusion"] = ("splitting every player count's range keeps every rank busy for the whole step: the cost is one drain tail (0.2 - 0.9 ms) and one launch "
                     "group per player count and rank, 1 - 3 % at eight ranks; dealing whole player counts leaves ranks idle behind the longest count "
                     "(k = 12: 559 of 2 200 ms): the range split stays")
if len(sys.argv) > 1:
    Path(sys.argv[1]).write_text(json.dumps(out, indent=1) + "\n")
