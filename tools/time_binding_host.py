"""Host cost of the engine binding per shuffle (build container only: imports the reference through oracle/ref_import.py).

TournamentBinding asks the engine for one tally per shuffle (shuffles_per_batch = 1) and hands every shuffle's result to the reference's own
chunk body as the reference's types (OutcomeCounter, defaultdict sums).  With a NULL engine (pre-made tallies, no simulation) what is left is
the binding's Python (tally_to_counters) + the reference's chunk body (_run_chunk / _run_chunk_metrics, run_tournament.py:403-585):
the ceiling of the binding route in shuffles per second, whatever the GPU does.
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
        tally = np.zeros((n, S, 26), dtype=np.int64)
        tally[:, :, 1] = 1
        tally[:, :, 2] = 1
        tally[:, : S // k, 0] = 1
        tally[:, : S // k, 4:15] = 7
        tally[:, : S // k, 15:26] = 49
        return {"tally": tally, "rows": None}


def measure(S_label: str, grid_kwargs: dict, k: int, n_shuffles: int) -> dict:
    strategies, _ = generate_strategy_grid(**grid_kwargs)
    S = len(strategies)
    cfg = rt.TournamentConfig(n_players=k, num_shuffles=n_shuffles, n_strategies=S)
    rt._init_worker(strategies, cfg, None, None)
    tasks = [rt.ShuffleTask(root_seed=7, k=k, shuffle_index=i, shuffle_seed=1000 + i, deterministic_batch_id=i // 30) for i in range(n_shuffles)]
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
    res = [measure("80", small, 2, 400), measure("5160", {}, 4, 40)]
    print(json.dumps(res, indent=1))
