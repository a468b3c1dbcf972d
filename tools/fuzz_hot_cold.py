"""One-off wider fuzz of the hot / cold kernel's instances against the oracle (the committed suite runs 60 fixed trials on the auto
plan): random legal tables, k = 3 .. 8, targets, round limits up to 400 (beyond the farkle field: replays), overrides, every
option combination.  usage: python tools/fuzz_hot_cold.py [trials=300] [seed=7]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np
import pyoracle as po
from farkle_ii_amd.backend import Engine, make_overrides
from farkle_ii_amd.strategies import STRATEGY_DTYPE

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rs = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
eng = Engine(0)
shapes = {}
for trial in range(trials):
    k = int(rs.integers(3, 9))
    S = k * int(rs.integers(2, 14))
    table = np.zeros(S, dtype=STRATEGY_DTYPE)
    for i in range(S):
        sf = int(rs.integers(0, 2)); so = int(rs.integers(0, 2)) if sf else 0
        cs, cd = int(rs.integers(0, 2)), int(rs.integers(0, 2))
        rb = int(rs.integers(0, 2)) if (cs and cd) else 0
        table[i] = (int(rs.choice([0, 1, 49, 50, 51, 199, 250, 300, 500, 1000, 1001, 1350, 10_000, 2_000_000])), int(rs.integers(-1, 7)), sf, so, cs, cd, rb,
                    int(rs.integers(0, 2)), int(rs.integers(0, 2)), int(rs.integers(0, 2)), 500 + i)
    target = int(rs.choice([49, 100, 500, 1_234, 2000, 9_999, 10_000, 10_001, 20_000, 135_000]))
    max_rounds = int(rs.choice([0, 1, 3, 50, 200, 255, 256, 400]))
    n_sh = int(rs.choice([1, 3, 9, 40]))
    root, first = int(rs.integers(0, 2**63)), int(rs.integers(0, 2**40))
    gps = S // k
    ovs = [(root, int(rs.integers(0, n_sh)), int(rs.integers(0, gps)), k, int(rs.choice([0, 2, 100, 300]))) for _ in range(int(rs.integers(0, 3)))]
    ovs = sorted({(o[1], o[2]): o for o in ovs}.values(), key=lambda o: (o[1], o[2]))
    ovs = [(o[0], first + o[1], o[2], o[3], o[4]) for o in ovs]
    opts = dict(hot_cold=int(rs.choice([-1, 1, 1])), hot_cold_lds=int(rs.choice([-1, 0, 1])), hot_cold_inc_regs=int(rs.choice([0, 1, 1])),
                hot_cold_tables=int(rs.choice([0, 1, 1])), hot_cold_waves=int(rs.choice([3, 5])), hot_cold_block=int(rs.choice([256, 256, 768, 1024])))
    for name, value in opts.items():
        eng.set_option(name, value)
    ref = po.tournament(table.view(po.STRATEGY_DTYPE), k, root, first, first + n_sh, shuffles_per_batch=3, target_score=target, max_rounds=max_rounds,
                        overrides=po.make_overrides(ovs) if ovs else None, want_rows=True, n_threads=8)
    got = eng.tournament(table, k, root, first, first + n_sh, shuffles_per_batch=3, target_score=target, max_rounds=max_rounds,
                         overrides=make_overrides(ovs) if ovs else None, want_rows=True, want_seat_stats=bool(trial % 2))
    t = eng.timing()
    shapes[(k, t["play_block"], t["play_lds_bytes"])] = shapes.get((k, t["play_block"], t["play_lds_bytes"]), 0) + 1
    ctx = (trial, k, S, target, max_rounds, n_sh, ovs, opts)
    assert np.array_equal(got["tally"], ref["tally"]), ctx
    assert got["rows"].tobytes() == ref["rows"].tobytes(), ctx
    if trial % 50 == 49:
        print(f"{trial + 1} trials ok", flush=True)
print("launch shapes seen (k, block, lds): " + ", ".join(f"{key}x{n}" for key, n in sorted(shapes.items())))
print(f"fuzz ok: {trials} trials")
