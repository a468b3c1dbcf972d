"""Diagnostic: distribution of game length (total rolls) on BASELINE config 2 and which pairings are the long ones."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from bench import grid64
from farkle_ii_amd.backend import Engine

table = grid64()
eng = Engine(0)
n_sh = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rows = eng.tournament(table, 2, 42, 0, n_sh, want_rows=True)["rows"]
rolls = rows["seats"]["rolls"].astype(np.int64).sum(axis=1)
status = rows["hdr"]["status"] if "hdr" in rows.dtype.names else rows["status"]
print("games", len(rolls), "mean rolls", rolls.mean(), "safety", (status != 0).mean())
for q in (50, 90, 99, 99.9, 99.99, 99.999, 100):
    print(f"p{q}: {np.percentile(rolls, q):.0f}")
comp = status == 0
print("completed only: ", [f"p{q}: {np.percentile(rolls[comp], q):.0f}" for q in (50, 99, 99.9, 99.99, 100)])
strat = rows["seats"]["strategy"]
# mean length by (has a never-banking-ish seat): classify by table fields
dthr = table["dice_threshold"][strat]; rb = table["require_both"][strat]
cls = ((dthr == 0) & (rb == 1)).sum(axis=1)   # seats with dice_thr 0 and AND rule (never bank)
for c in (0, 1, 2):
    m = cls == c
    if m.any(): print(f"never-bank seats = {c}: share {m.mean():.4f} mean rolls {rolls[m].mean():.1f} p99.9 {np.percentile(rolls[m], 99.9):.0f} max {rolls[m].max()}")
# per-strategy mean game length (either seat)
S = len(table)
tot = np.zeros(S); cnt = np.zeros(S)
for s in range(2):
    np.add.at(tot, strat[:, s], rolls); np.add.at(cnt, strat[:, s], 1)
order = np.argsort(-tot / cnt)
for i in order[:12]:
    t = table[i]
    print(f"strategy {i}: mean game rolls {tot[i]/cnt[i]:.0f}  thr {t['score_threshold']} dthr {t['dice_threshold']} rb {t['require_both']} fav {t['favor_score']}")
