"""One-off wide fuzz of the SHIPPED launch plan against the oracle (round 5: fused discard key, 32-bit score entries in both table forms,
buffered half word in the cold slot, instances for nine to twelve seats, per-k hand-over thresholds): random legal tables, k = 2 .. 12,
targets (units-of-50 rounding), round limits incl. 0 and beyond the farkle field (replays), overrides, and the options a caller can
set — max_waves, batch_threshold, use_lds_tally, hot_cold, chunk_bytes, pipeline.  Tallies, rows, all-seat statistics and (every
third trial) the float64 ratio sums.  usage: python tools/fuzz_shipped_plan.py [trials=300] [seed=11]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np
import pyoracle as po
from farkle_ii_amd.backend import Engine, make_overrides
from farkle_ii_amd.strategies import STRATEGY_DTYPE
from oracle_engine_stub import seat_ratio_sums_from_rows, seat_stats_from_rows

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
eng = Engine(0)
shapes = {}
defaults = dict(max_waves=6, batch_threshold=0, use_lds_tally=-1, hot_cold=-1, chunk_bytes=48 << 30, pipeline=1)
try:
    for trial in range(trials):
        k = int(rs.integers(2, 13))
        S = k * int(rs.integers(2, 14))
        table = np.zeros(S, dtype=STRATEGY_DTYPE)
        for i in range(S):
            sf = int(rs.integers(0, 2)); so = int(rs.integers(0, 2)) if sf else 0
            cs, cd = int(rs.integers(0, 2)), int(rs.integers(0, 2))
            rb = int(rs.integers(0, 2)) if (cs and cd) else 0
            table[i] = (int(rs.choice([0, 1, 49, 50, 51, 199, 250, 300, 500, 1000, 1001, 1350, 10_000, 2_000_000])), int(rs.integers(-1, 7)), sf, so, cs, cd, rb,
                        int(rs.integers(0, 2)), int(rs.integers(0, 2)), int(rs.integers(0, 2)), 500 + i)
        if trial % 5 == 4:  # tables whose strategies share every flag: the scalar-flag instances
            for name in ("smart_five", "smart_one", "consider_score", "consider_dice", "require_both", "auto_hot_dice", "run_up_score", "favor_score"):
                table[name] = table[name][0]
        target = int(rs.choice([49, 100, 500, 1_234, 2000, 9_999, 10_000, 10_001, 20_000, 135_000]))
        max_rounds = int(rs.choice([0, 1, 3, 50, 200, 255, 256, 400]))
        n_sh = int(rs.choice([1, 3, 9, 40]))
        spb = int(rs.choice([1, 3, 16]))
        root, first = int(rs.integers(0, 2**63)), int(rs.integers(0, 2**40))
        gps = S // k
        ovs = [(root, int(rs.integers(0, n_sh)), int(rs.integers(0, gps)), k, int(rs.choice([0, 2, 100, 300]))) for _ in range(int(rs.integers(0, 3)))]
        ovs = sorted({(o[1], o[2]): o for o in ovs}.values(), key=lambda o: (o[1], o[2]))
        ovs = [(o[0], first + o[1], o[2], o[3], o[4]) for o in ovs]
        opts = dict(max_waves=int(rs.choice([6, 6, 4, 3])), batch_threshold=int(rs.choice([0, 0, 1, 8, 64])), use_lds_tally=int(rs.choice([-1, -1, 0])),
                    hot_cold=int(rs.choice([-1, -1, -1, 0])), chunk_bytes=int(rs.choice([48 << 30, 48 << 30, 1 << 20])), pipeline=int(rs.choice([1, 1, 0])))
        for name, value in opts.items():
            eng.set_option(name, value)
        ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, root, first, first + n_sh, shuffles_per_batch=spb, target_score=target, max_rounds=max_rounds,
                            overrides=po.make_overrides(ovs) if ovs else None, want_rows=True, n_threads=8)
        got = eng.tournament(table, k, root, first, first + n_sh, shuffles_per_batch=spb, target_score=target, max_rounds=max_rounds,
                             overrides=make_overrides(ovs) if ovs else None, want_rows=True, want_seat_stats=bool(trial % 2),
                             want_seat_ratios=trial % 3 == 0)
        t = eng.timing()
        key = (k, t["play_block"], t["play_lds_bytes"], t["play_mixed_flags"])
        shapes[key] = shapes.get(key, 0) + 1
        ctx = (trial, k, S, target, max_rounds, n_sh, spb, ovs, opts)
        assert np.array_equal(got["tally"], ref["tally"]), ctx
        assert got["rows"].tobytes() == ref["rows"].tobytes(), ctx
        if trial % 2:
            assert np.array_equal(got["seat_stats"], seat_stats_from_rows(ref["rows"], k, S, gps, spb)), ctx
            if trial % 3 == 0:
                assert got["seat_ratio_sums"].tobytes() == seat_ratio_sums_from_rows(ref["rows"], k, S, gps, spb).tobytes(), ctx
        counts = eng.tournament(table, k, root, first, first + n_sh, target_score=target, max_rounds=max_rounds,
                                overrides=make_overrides(ovs) if ovs else None)  # counts only: lean state records, LDS tally where it fits
        assert np.array_equal(counts["tally"][0], ref["tally"].sum(axis=0)), ctx
        if trial % 50 == 49:
            print(f"{trial + 1} trials ok", flush=True)
finally:
    for name, value in defaults.items():
        eng.set_option(name, value)
print("launch shapes seen (k, block, lds, flag form): " + ", ".join(f"{key}x{n}" for key, n in sorted(shapes.items())))
print(f"fuzz ok: {trials} trials")
