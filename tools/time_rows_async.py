"""Diagnostic (GPU): eight 256-MB launch groups of column images back to back into two page-locked buffers — counts only, the waiting rows
call, and the async form (option "rows_async": the call returns when its last copy is queued; fk_rows_wait) — per call and in total.
usage: python tools/time_rows_async.py"""
import sys, time, numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from farkle_ii_amd.backend import Engine, row_columns_bytes
from farkle_ii_amd.strategies import generate_strategy_grid, pack_strategies, prepare_public_helper_strategies
table = pack_strategies(prepare_public_helper_strategies(generate_strategy_grid()[0]))
ids = np.arange(len(table), dtype=np.int32)
eng = Engine(0)
for k, n_sh in ((2, 800), (5, 900), (12, 930)):
    gps = len(table) // k
    stride = row_columns_bytes(k, gps)
    pins = [eng.pinned_empty(n_sh * stride, np.uint8) for _ in range(2)]
    G = 8
    def run(async_rows, hint):
        t0 = time.perf_counter(); per = []
        prev = None
        for g in range(G):
            if hint and g + 1 < G:
                eng.hint_next((g + 1) * n_sh, (g + 2) * n_sh, need_state=True)
            t1 = time.perf_counter()
            res = eng.tournament_columns(table, k, 102, g * n_sh, (g + 1) * n_sh, ids, columns_out=pins[g & 1], async_rows=async_rows)
            per.append((time.perf_counter() - t1) * 1e3)
            if async_rows:
                if prev is not None: eng.rows_wait(prev)
                prev = res["rows_event"]
        if async_rows: eng.rows_wait(prev)
        return (time.perf_counter() - t0) * 1e3, per
    def counts(hint):
        t0 = time.perf_counter()
        for g in range(G):
            if hint and g + 1 < G:
                eng.hint_next((g + 1) * n_sh, (g + 2) * n_sh, need_state=False)
            eng.tournament(table, k, 102, g * n_sh, (g + 1) * n_sh)
        return (time.perf_counter() - t0) * 1e3
    for hint in (False, True):
        counts(hint); run(False, hint); run(True, hint)
        c = min(counts(hint) for _ in range(3))
        s = min((run(False, hint) for _ in range(3)), key=lambda x: x[0])
        a = min((run(True, hint) for _ in range(3)), key=lambda x: x[0])
        print(f"k={k} hint={hint} {G} groups of {n_sh*stride/1e6:.0f} MB: counts {c:.1f} ms; waiting {s[0]:.1f} ms; async {a[0]:.1f} ms; per call waiting {[round(x,1) for x in s[1]]} async {[round(x,1) for x in a[1]]}", flush=True)
