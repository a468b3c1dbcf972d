"""Test helpers: a synthetic H2H schedule (block dicts + strategy manifest frame) and the reference's serial custom-runner
loop (execute_h2h_schedule, src/farkle/analysis/h2h_schedule.py:2038-2093, with plan_h2h_chunk :94-129) restated."""
from __future__ import annotations

import numpy as np


def manifest_frame(table: np.ndarray):
    import pandas as pd

    return pd.DataFrame([{"strategy_id": int(r["strategy_id"]), "score_threshold": int(r["score_threshold"]),
                          "dice_threshold": int(r["dice_threshold"]), "smart_five": bool(r["smart_five"]), "smart_one": bool(r["smart_one"]),
                          "consider_score": bool(r["consider_score"]), "consider_dice": bool(r["consider_dice"]),
                          "require_both": bool(r["require_both"]), "auto_hot_dice": bool(r["auto_hot_dice"]),
                          "run_up_score": bool(r["run_up_score"]), "favor_dice_or_score": "score" if r["favor_score"] else "dice"}
                         for r in table])


def make_schedule(table: np.ndarray, n_blocks: int, seed: int, roots=(42, 43), target_range=(3, 40), multiplier: float = 1.5):
    """n_blocks (pair, root, order) blocks over random strategy pairs of `table`, in the reference's submission order."""
    rng = np.random.default_rng(seed)
    ids = table["strategy_id"].astype(int)
    blocks = []
    pair_id = 0
    while len(blocks) < n_blocks:
        a, b = (int(x) for x in rng.choice(ids, 2, replace=False))
        target = int(rng.integers(*target_range))
        for ri, root in enumerate(roots):
            for order in (0, 1):
                s1, s2 = (a, b) if order == 0 else (b, a)
                blocks.append({"block_id": f"p{pair_id}-r{ri}-o{order}", "family_hash": "fam", "schedule_hash": "sch", "pair_id": pair_id,
                               "root_index": ri, "root_seed": int(root), "order": order, "strategy_a": a, "strategy_b": b,
                               "seat1_strategy": s1, "seat2_strategy": s2, "n_completed_required": target,
                               "max_attempts": int(np.ceil(multiplier * target)), "rng_scheme_version": 2, "rng_purpose_namespace": 203})
        pair_id += 1
    return blocks[:n_blocks]


def serial_schedule_loop(blocks, runner, manifest_path, chunk_games: int):
    """execute_h2h_schedule's custom-runner branch: one block at a time, chunk by chunk, until its terminal status."""
    out, calls = [], 0
    for block in blocks:
        current = dict(block)
        while True:
            attempted = int(current.get("games_attempted", 0))
            count = min(int(current["max_attempts"]), attempted + chunk_games) - attempted
            assert count > 0
            result = runner(current, manifest_path, count)
            calls += 1
            for field in ("block_id", "family_hash", "schedule_hash"):
                assert str(result[field]) == str(block[field])
            assert result["games_attempted"] == result["games_completed"] + result["games_safety_limit"]
            assert result["wins_seat1"] + result["wins_seat2"] == result["games_completed"] <= block["n_completed_required"]
            assert attempted < result["games_attempted"] <= attempted + count
            if result["completion_status"] != "partial_resumable":
                out.append(result)
                break
            current = result
    return out, calls
