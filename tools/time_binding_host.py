"""Host cost of the engine binding per shuffle (build container only: imports the reference through oracle/ref_import.py).

Round 6: TournamentBinding serves `_run_chunk` and `_run_chunk_metrics` (without row shards) per CHUNK — one launch with ONE tally for the
deterministic batch, converted once to the reference's types (OutcomeCounter, defaultdict sums), one progress event.  (Round 5 asked for a
tally per shuffle and ran the reference's per-shuffle loop: 0.17 - 8 ms of Python per shuffle, profiles/r05_binding_host_cost.json.)  With a
NULL engine (pre-made tallies, no simulation) what is left is the binding's Python: the ceiling of the binding route, whatever the GPU does.
    python tools/time_binding_host.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import ref_import

ref_import.import_reference()
from farkle.simulation import run_tournament as rt  # noqa: E402
from farkle.simulation.simulation import generate_strategy_grid  # noqa: E402

from farkle_ii_amd.reference_binding import TournamentBinding, pack_reference_strategies  # noqa: E402


class NullEngine:
    """Returns a plausible tally per shuffle without playing: every game completed, seat 0's strategy wins."""

    def __init__(self):
        self.calls = 0

    def tournament(self, table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=None, **kw):
        self.calls += 1
        n, S = shuffle_end - shuffle_begin, len(table)
        spb = shuffles_per_batch or n
        nb = (n + spb - 1) // spb
        per = np.minimum(spb, n - np.arange(nb) * spb)[:, None]
        tally = np.zeros((nb, S, 26), dtype=np.int64)
        tally[:, :, 1] = per
        tally[:, :, 2] = per
        tally[:, : S // k, 0] = per
        tally[:, : S // k, 4:15] = 7 * per
        tally[:, : S // k, 15:26] = 49 * per
        return {"tally": tally, "rows": None}


def measure(S_label: str, grid_kwargs: dict, k: int, n_shuffles: int) -> dict:
    strategies, _ = generate_strategy_grid(**grid_kwargs)
    S = len(strategies)
    cfg = rt.TournamentConfig(n_players=k, num_shuffles=n_shuffles, n_strategies=S)
    rt._init_worker(strategies, cfg, None, None)
    tasks = [rt.ShuffleTask(root_seed=7, k=k, shuffle_index=i, shuffle_seed=1000 + i, deterministic_batch_id=0) for i in range(n_shuffles)]  # one deterministic batch
    eng = NullEngine()
    out = {"S": S, "k": k, "shuffles": n_shuffles}
    with TournamentBinding(rt, engine=eng) as b:
        for name, call in (("_run_chunk", lambda: rt._run_chunk(tasks)), ("_run_chunk_metrics", lambda: rt._run_chunk_metrics(tasks))):
            call()
            t0 = time.perf_counter()
            call()
            dt = time.perf_counter() - t0
            out[name] = {"ms_per_shuffle": dt / n_shuffles * 1e3, "shuffles_per_s": n_shuffles / dt, "games_per_s": n_shuffles * (S // k) / dt}
    return out


if __name__ == "__main__":
    small = dict(score_thresholds=[200, 250, 300, 350, 400], dice_thresholds=[0, 1, 2, 3], smart_five_opts=[True], smart_one_opts=[True],
                 consider_score_opts=[True], consider_dice_opts=[True], auto_hot_dice_opts=[True], run_up_score_opts=[True])
    res = [measure("80", small, 2, 400), measure("5160", {}, 4, 43)]  # (43 shuffles: a deterministic batch of the production plan)
    before = json.loads((ROOT / "profiles" / "r05_binding_host_cost.json").read_text())
    for now, old in zip(res, before):
        for name in ("_run_chunk", "_run_chunk_metrics"):
            now[name]["speedup_vs_round_5_per_shuffle_service"] = now[name]["games_per_s"] / old[name]["games_per_s"]
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(res, indent=1) + "\n")
