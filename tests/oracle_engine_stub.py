"""TEST INFRASTRUCTURE ONLY — an `Engine`-shaped object backed by the CPU oracle.

It exists so that the multi-rank plumbing of ``bench.py`` / ``farkle run`` (rank launch, shuffle-range shards, the one
tally reduce, rank-0 artifacts) can be exercised on GPU-less hosts with the gloo backend.  Nothing in the product imports
it; ``bench.py`` loads it only through the explicit ``FK_BENCH_ENGINE=oracle_engine_stub:Engine`` test hook and then
labels its output as a stub run."""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

import pyoracle as po  # noqa: E402


def seat_stats_from_rows(rows: np.ndarray, k: int, S: int, gps: int, spb: int) -> np.ndarray:
    """The integer accumulators of fk_tournament_run_stats computed from rows, the way the reference's
    analysis/all_player_metrics.py:257-340 sums them (one exposure per seat; rank / loss_margin over completed games)."""
    n = len(rows)
    batch = (np.arange(n) // gps) // max(spb, 1)
    out = np.zeros((int(batch.max()) + 1 if n else 1, S, 31), dtype=np.int64)
    completed = rows["status"] == 0
    winning = rows["seats"]["score"].astype(np.int64).max(axis=1)
    rounds = rows["n_rounds"].astype(np.int64)
    for seat in range(k):
        x = rows["seats"][:, seat]
        strat = x["strategy"].astype(np.int64)
        score, turns = x["score"].astype(np.int64), x["n_turns"].astype(np.int64)
        tmr = turns - rounds
        rank = x["rank"].astype(np.int64)
        margin = np.where(completed, winning - score, 0)
        cols = {0: np.ones(n, dtype=np.int64), 1: completed.astype(np.int64), 2: (~completed).astype(np.int64),
                3: (completed & (rows["winner_seat"] == seat)).astype(np.int64), 4: score, 5: score * score, 6: turns, 7: turns * turns,
                8: (tmr != 0).astype(np.int64), 9: tmr, 10: tmr * tmr, 11: np.where(completed, rank, 0),
                12: np.where(completed, rank * rank, 0), 13: margin, 14: margin * margin}
        for j, name in enumerate(("rolls", "farkles", "highest_turn", "hot_dice", "smart_five_uses", "n_smart_five_dice",
                                  "smart_one_uses", "n_smart_one_dice")):
            v = x[name].astype(np.int64)
            cols[15 + 2 * j], cols[16 + 2 * j] = v, v * v
        for c, v in cols.items():
            np.add.at(out[:, :, c], (batch, strat), v)
    return out


def seat_ratio_sums_from_rows(rows: np.ndarray, k: int, S: int, gps: int, spb: int) -> np.ndarray:
    """The four float64 accumulators of fk_tournament_run_all_player from rows, the way the reference adds them
    (analysis/all_player_metrics.py:308-321 + np.add.at :174-177): exposures flattened in (row, seat) order, unbuffered."""
    n = len(rows)
    batch = (np.arange(n) // gps) // max(spb, 1)
    out = np.zeros((int(batch.max()) + 1 if n else 1, S, 4), dtype=np.float64)
    rounds = rows["n_rounds"].astype(np.float64)
    strat = np.stack([rows["seats"][:, seat]["strategy"].astype(np.int64) for seat in range(k)], axis=1).reshape(-1)
    b = np.repeat(batch, k)
    cols = [[], [], [], []]
    for seat in range(k):
        x = rows["seats"][:, seat]
        score, turns = x["score"].astype(np.float64), x["n_turns"].astype(np.float64)
        exact = np.divide(score, turns, out=np.zeros_like(score), where=turns != 0)
        proxy = np.divide(score, rounds, out=np.zeros_like(score), where=rounds != 0)
        for c, v in enumerate((exact, exact * exact, proxy, proxy * proxy)):
            cols[c].append(v)
    for c in range(4):
        np.add.at(out[:, :, c], (b, strat), np.stack(cols[c], axis=1).reshape(-1))
    return out


def column_images(rows: np.ndarray, k: int, ids: np.ndarray, gps: int) -> np.ndarray:
    """Device rows (AoS) -> the per-shuffle column images of include/farkle_hip.h (fk_tournament_run_columns), with NumPy: int32 planes
    [4 + 13 k][gps] in the row schema's column order, then status / winner_seat / rank_order bytes.  Null fields hold 0."""
    from farkle_ii_amd.backend import row_columns_bytes

    n = len(rows)
    n_sh = n // gps if gps else 0
    out = np.zeros((n_sh, row_columns_bytes(k, gps)), dtype=np.uint8)
    if n == 0:
        return out
    seats = rows["seats"]
    completed = rows["status"] == 0
    w = np.where(completed, rows["winner_seat"].astype(np.int64), 0)
    ar = np.arange(n)
    scores = seats["score"].astype(np.int64)
    win = scores[ar, w]
    others = scores.copy()
    others[ar, w] = np.iinfo(np.int64).min
    second = others.max(axis=1) if k > 1 else np.zeros(n, dtype=np.int64)
    planes = np.zeros((4 + 13 * k, n), dtype=np.int32)
    planes[0] = np.where(completed, ids[seats["strategy"][ar, w]], 0)
    planes[1] = np.where(completed, win, 0)
    planes[2] = np.where(completed, win - second, 0)
    planes[3] = rows["n_rounds"]
    for s in range(k):
        x, b = seats[:, s], 4 + 13 * s
        for f, name in enumerate(("score", "farkles", "rolls", "highest_turn")):
            planes[b + f] = x[name]
        planes[b + 4] = ids[x["strategy"]]
        planes[b + 5] = np.where(completed, x["rank"], 0)
        planes[b + 6] = np.where(completed, win - scores[:, s], 0)
        for f, name in enumerate(("smart_five_uses", "n_smart_five_dice", "smart_one_uses", "n_smart_one_dice", "hot_dice", "n_turns")):
            planes[b + 7 + f] = x[name]
    order = np.where(completed[:, None], np.argsort(seats["rank"].astype(np.int64), axis=1, kind="stable"), 0).astype(np.uint8)
    ni = (4 + 13 * k) * 4 * gps
    for i in range(n_sh):
        sl = slice(i * gps, (i + 1) * gps)
        out[i, :ni] = np.ascontiguousarray(planes[:, sl]).view(np.uint8).reshape(-1)
        out[i, ni:ni + gps] = (~completed[sl]).astype(np.uint8)
        out[i, ni + gps:ni + 2 * gps] = w[sl].astype(np.uint8)
        out[i, ni + 2 * gps:ni + 2 * gps + gps * k] = order[sl].reshape(-1)
    return out


class Engine:
    def __init__(self, device: int = 0):
        self.device = device
        self._games = 0

    def close(self) -> None:
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        pass

    def set_option(self, name: str, value: int) -> None:
        pass

    def hint_next(self, shuffle_begin: int, shuffle_end: int, need_state: bool = False) -> None:
        pass

    def device_info(self) -> dict:
        return {"name": "CPU oracle stub", "arch": "host", "compute_units": 1, "clock_mhz": 1, "wavefront_size": 1,
                "lds_bytes_per_cu": 0, "hbm_bytes": 0}

    def timing(self) -> dict:
        return {"perm_ms": 0.0, "seed_ms": 0.0, "play_ms": 1.0, "total_ms": 1.0, "play_launches": 1, "play_block": 0,
                "play_grid": 0, "play_lds_bytes": 0, "games": self._games}

    def tournament(self, table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=None, target_score=10_000,
                   max_rounds=200, overrides=None, want_rows=False, want_perms=False, want_seat_stats=False, want_seat_ratios=True) -> dict:
        t = np.ascontiguousarray(table).view(po.STRATEGY_DTYPE)
        ov = None if overrides is None or len(overrides) == 0 else np.ascontiguousarray(overrides).view(po.OVERRIDE_DTYPE)
        res = po.tournament(t, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=shuffles_per_batch,
                            target_score=target_score, max_rounds=max_rounds, overrides=ov, want_rows=want_rows or want_seat_stats,
                            want_perms=want_perms, n_threads=2)
        self._games = (shuffle_end - shuffle_begin) * (len(t) // k)
        stats = ratios = None
        if want_seat_stats:
            spb = (shuffle_end - shuffle_begin) if not shuffles_per_batch else shuffles_per_batch
            stats = seat_stats_from_rows(res["rows"], k, len(t), len(t) // k, spb)
            ratios = seat_ratio_sums_from_rows(res["rows"], k, len(t), len(t) // k, spb) if want_seat_ratios else None
        return {"tally": res["tally"], "rows": res["rows"] if want_rows else None, "perms": res["perms"], "seat_stats": stats,
                "seat_ratio_sums": ratios}

    def tournament_columns(self, table, k, root_seed, shuffle_begin, shuffle_end, strategy_ids, shuffles_per_batch=None, target_score=10_000,
                           max_rounds=200, overrides=None, columns_out=None, async_rows=False, shuffle_seeds_out=None,
                           game_seeds_out=None) -> dict:
        """fk_tournament_run_columns from the oracle's ROWS: the per-shuffle column images, restated with NumPy."""
        res = self.tournament(table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=shuffles_per_batch,
                              target_score=target_score, max_rounds=max_rounds, overrides=overrides, want_rows=True)
        images = column_images(res["rows"], k, np.asarray(strategy_ids, dtype=np.int32), len(table) // k)
        if columns_out is not None:  # the caller's buffer (farkle run: a ring of image buffers, reused while earlier groups are still written)
            out = columns_out.reshape(-1)[:images.size].reshape(images.shape)
            out[...] = images
            images = out
        from farkle_ii_amd import random as urandom

        sh = np.arange(shuffle_begin, shuffle_end, dtype=np.uint64)
        gps = len(table) // k
        if shuffle_seeds_out is not None:  # the ns-100 / ns-102 fingerprints, with NumPy (farkle_ii_amd.random restates utils/random.py)
            shuffle_seeds_out[...] = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_SHUFFLE, root_seed=root_seed, k=k, shuffle_index=sh,
                                                              dtype=np.uint32)
        if game_seeds_out is not None and len(sh):
            game_seeds_out[...] = urandom.coordinate_seeds(urandom.RandomPurpose.TOURNAMENT_GAME, root_seed=root_seed, k=k,
                                                           shuffle_index=np.repeat(sh, gps), game_index=np.tile(np.arange(gps, dtype=np.uint64), len(sh)),
                                                           dtype=np.uint32)
        self._rows_calls = getattr(self, "_rows_calls", 0) + (1 if async_rows else 0)
        return {"tally": res["tally"], "columns": images, "rows_event": (self._rows_calls - 1) % 4 if async_rows else None}

    columns_with_seeds = True

    def rows_wait(self, slot: int) -> None:  # (the images are complete when tournament_columns returns; farkle run pipelines its groups
        assert 0 <= slot < 4                 # through a launcher thread when the engine has this method)

    def pinned_empty(self, n: int, dtype) -> np.ndarray:
        return np.full(int(n), 0xA5, dtype=dtype)  # (stale bytes of an earlier group, never zeros)

    def tournament_lags(self, table, k, root_seed, shuffle_begin, shuffle_end, lags, shuffles_per_batch=None, target_score=10_000,
                        max_rounds=200, overrides=None) -> dict:
        """fk_tournament_run_lags from ROWS: the value matrix (n_rounds | won << 15) per (shuffle, strategy), then the host
        statement of the rule (rng_lags.LagSummary.from_series)."""
        from farkle_ii_amd.rng_lags import LagSummary

        res = self.tournament(table, k, root_seed, shuffle_begin, shuffle_end, shuffles_per_batch=shuffles_per_batch,
                              target_score=target_score, max_rounds=max_rounds, overrides=overrides, want_rows=True)
        rows, S = res["rows"], len(table)
        n_sh, gps = shuffle_end - shuffle_begin, len(table) // k
        values = np.zeros((max(n_sh, 0), S), dtype=np.uint16)
        if n_sh > 0:
            sh = np.arange(len(rows)) // gps
            completed = rows["status"] == 0
            for seat in range(k):
                won = completed & (rows["winner_seat"] == seat)
                values[sh, rows["seats"][:, seat]["strategy"]] = (rows["n_rounds"] & 0x7FFF) | (won.astype(np.uint16) << 15)
        summ = LagSummary.from_series(values, lags) if n_sh > 0 else None
        m = min(max(lags), max(n_sh, 0))
        return {"tally": res["tally"], "lag_sums": summ.sums if summ else np.zeros((S, len(lags), 11), dtype=np.int64),
                "lag_head": values[:m], "lag_tail": values[n_sh - m:] if n_sh > 0 else values[:0], "n_shuffles": max(n_sh, 0)}

    def play_games(self, coords, table, seat_strategy, k, target_score=10_000, max_rounds=200):
        return po.play_games(np.ascontiguousarray(coords).view(po.COORD_DTYPE), np.ascontiguousarray(table).view(po.STRATEGY_DTYPE),
                             seat_strategy, k, target_score=target_score, max_rounds=max_rounds, n_threads=2)

    def h2h(self, seats, root_seed, pair_id, order, target, max_attempts, chunk_games, target_score=10_000, max_rounds=200,
            overrides=None, state=None):
        ov = None if overrides is None or len(overrides) == 0 else np.ascontiguousarray(overrides).view(po.OVERRIDE_DTYPE)
        return po.h2h_block(np.ascontiguousarray(seats).view(po.STRATEGY_DTYPE), root_seed, pair_id, order, target, max_attempts,
                            chunk_games, target_score=target_score, max_rounds=max_rounds, overrides=ov, state=state)

    def h2h_blocks(self, seats, root_seed, pair_ids, orders, target, max_attempts, chunk_games=None, target_score=10_000,
                   max_rounds=200, overrides=None, states=None):
        seats = np.ascontiguousarray(seats).reshape(-1, 2)
        n = len(seats)
        target = np.broadcast_to(np.asarray(target, dtype=np.uint64), (n,))
        max_attempts = np.broadcast_to(np.asarray(max_attempts, dtype=np.uint64), (n,))
        chunk = int(max_attempts.max()) if chunk_games is None and n else int(chunk_games or 0)
        out = np.zeros((n, 5), dtype=np.uint64)
        for b in range(n):
            out[b] = self.h2h(seats[b], root_seed, int(pair_ids[b]), int(orders[b]), int(target[b]), int(max_attempts[b]), chunk,
                              target_score=target_score, max_rounds=max_rounds, overrides=overrides,
                              state=None if states is None else np.asarray(states, dtype=np.uint64).reshape(n, 5)[b])
            self._games = int(out[b][0])
        return out
