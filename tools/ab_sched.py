"""A/B (round 6): the same kernel sources under other LLVM machine-scheduler strategies.  profiles/r06_cndmask_run_length.txt shows that the
"half-rate" instruction class only costs 4.2 cycles in PURE streams — mixed 1 : 1 with full-rate instructions it is hidden — so the ORDER the
compiler emits the roll loop in matters, and the scheduler strategy is the one knob that changes it without touching the source.
Builds libfarkle_hip_<name>.so per strategy (backend.VARIANTS), checks every tally against the product library's, times the game kernel
(HIP events, fk_timing.play_ms) on BASELINE config 2 (k = 2, 64 strategies, 10^7 games) and on k = 8 / 5 160 strategies (2.6 x 10^7 games).
usage: python tools/ab_sched.py [out.json]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
from farkle_ii_amd import backend
from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies

STRATEGIES = {"max-ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "max-memory-clause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
              # ("iterative-ilp" crashes clang 22 / ROCm 7.2 in the register allocator on fk_seat_ratio_kernel: not built)
              "iterative-minreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"]}
# second set (usage: python tools/ab_sched.py OUT.json codegen): other code-generation knobs of the same compiler over the same sources — what
# decides the copies (50 v_mov_b32 in the k = 8 roll loop), the branches (lane utilisation 0.64 - 0.80) and the order within a block
CODEGEN = {"early-ifcvt": ["-mllvm", "-amdgpu-early-ifcvt=1"], "no-vgpr-liverange-opt": ["-mllvm", "-amdgpu-opt-vgpr-liverange=0"],
           "no-unclustered-high-rp-reschedule": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1"],
           "no-clustered-low-occupancy-reschedule": ["-mllvm", "-amdgpu-disable-clustered-low-occupancy-reschedule=1"],
           "schedule-metric-bias-0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"], "schedule-metric-bias-100": ["-mllvm", "-amdgpu-schedule-metric-bias=100"],
           "no-loop-alignment": ["-mllvm", "-amdgpu-disable-loop-alignment=1"], "O2": ["-O2"]}
if len(sys.argv) > 2 and sys.argv[2] == "codegen":
    STRATEGIES = CODEGEN
for name, flags in STRATEGIES.items():
    backend.VARIANTS["sched_" + name.replace("-", "_")] = flags
g64, _ = generate_strategy_grid(score_thresholds=[250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True], smart_one_opts=[True],
                                consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True], run_up_score_opts=[True])
t64 = pack_strategies(g64)
t5160 = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))
CASES = {"config2_k2": (t64, 2, 42, 0, 312_500), "k8_5160": (t5160, 8, 0, 0, 40_000), "k5_5160": (t5160, 5, 0, 0, 25_000)}


def measure(variant):
    eng = backend.Engine(0, variant=variant)
    out = {}
    for case, (table, k, root, lo, hi) in CASES.items():
        eng.tournament(table, k, root, lo, lo + 2000)  # warm
        ms, tally = [], None
        for _ in range(4):
            tally = eng.tournament(table, k, root, lo, hi)["tally"]
            ms.append(eng.timing()["play_ms"])
        out[case] = {"play_ms_min": min(ms), "play_ms_all": ms, "tally_sha": __import__("hashlib").sha256(tally.tobytes()).hexdigest()}
    eng.close()
    return out


if __name__ == "__main__":
    res = {"product": measure(None)}
    print("product", {c: round(v["play_ms_min"], 3) for c, v in res["product"].items()}, flush=True)
    for name in STRATEGIES:
        variant = "sched_" + name.replace("-", "_")
        try:
            backend.build_library(variant=variant)
            r = measure(variant)
        except Exception as exc:  # noqa: BLE001
            res[name] = {"error": f"{type(exc).__name__}: {str(exc)[:300]}"}
            print(name, res[name], flush=True)
            continue
        for c, v in r.items():
            v["parity"] = v["tally_sha"] == res["product"][c]["tally_sha"]
            v["vs_product"] = res["product"][c]["play_ms_min"] / v["play_ms_min"]
        res[name] = r
        print(name, {c: (round(v["play_ms_min"], 3), round(v["vs_product"], 4), v["parity"]) for c, v in r.items()}, flush=True)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(res, indent=1) + "\n")
