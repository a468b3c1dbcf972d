"""Host-side strategy objects and grid enumeration.

Mirrors the reference's operator surface for the hot path's inputs
(``src/farkle/simulation/strategies.py``: ``ThresholdStrategy`` :165-290, ``iter_strategy_combos``
:346-396, ``_favor_options`` :335-343, ``build_stop_at_strategy`` :455-482,
``random_threshold_strategy`` :418-452; ``src/farkle/simulation/simulation.py:55-221``
``generate_strategy_grid``).  Strategies are plain data here: the decisions themselves are
evaluated on the GPU; :meth:`ThresholdStrategy.pack` produces the 20-byte ``fk_strategy`` record
of ``include/farkle_hip.h``.
"""
from __future__ import annotations

import copy as _copy

import re
from dataclasses import dataclass, replace
from enum import Enum
from typing import Any, Iterable, Iterator, Sequence

import numpy as np

__all__ = [
    "FavorDiceOrScore", "ThresholdStrategy", "StopAtStrategy", "STOP_AT_THRESHOLDS", "STRATEGY_TUPLE_FIELDS",
    "DEFAULT_STRATEGY_GRID", "iter_strategy_combos", "generate_strategy_grid", "default_grid_tuples",
    "build_stop_at_strategy", "random_threshold_strategy", "strategy_tuple", "pack_strategies",
    "STRATEGY_DTYPE", "prepare_public_helper_strategies", "experiment_size",
]


class FavorDiceOrScore(Enum):
    """Tie-break preference of the Smart-discard argmax (strategies.py:48-55)."""

    SCORE = "score"
    DICE = "dice"

    def __str__(self) -> str:
        return self.value


STOP_AT_THRESHOLDS: tuple[int, ...] = (350, 400, 450, 500)  # strategies.py:58

STRATEGY_TUPLE_FIELDS: tuple[str, ...] = (
    "score_threshold", "dice_threshold", "smart_five", "smart_one", "consider_score", "consider_dice",
    "require_both", "auto_hot_dice", "run_up_score", "favor_dice_or_score",
)

DEFAULT_STRATEGY_GRID: dict[str, tuple[Any, ...]] = {  # strategies.py:76-85
    "score_thresholds": tuple(range(200, 1400, 50)),
    "dice_thresholds": tuple(range(0, 5)),
    "smart_five_opts": (True, False),
    "smart_one_opts": (True, False),
    "consider_score_opts": (True, False),
    "consider_dice_opts": (True, False),
    "auto_hot_dice_opts": (False, True),
    "run_up_score_opts": (True, False),
}

# 20-byte device record, include/farkle_hip.h: fk_strategy
STRATEGY_DTYPE = np.dtype(
    [("score_threshold", "<i4"), ("dice_threshold", "<i4"), ("smart_five", "u1"), ("smart_one", "u1"),
     ("consider_score", "u1"), ("consider_dice", "u1"), ("require_both", "u1"), ("auto_hot_dice", "u1"),
     ("run_up_score", "u1"), ("favor_score", "u1"), ("strategy_id", "<i4")]
)


@dataclass
class ThresholdStrategy:
    """Threshold-based keep/bank rule (10 parameters + optional integer id)."""

    score_threshold: int = 300
    dice_threshold: int = 2
    smart_five: bool = False
    smart_one: bool = False
    consider_score: bool = True
    consider_dice: bool = True
    require_both: bool = False
    auto_hot_dice: bool = False
    run_up_score: bool = False
    favor_dice_or_score: FavorDiceOrScore = FavorDiceOrScore.SCORE
    strategy_id: int | None = None

    def __post_init__(self) -> None:
        if self.smart_one and not self.smart_five:
            raise ValueError("ThresholdStrategy: smart_one=True requires smart_five=True")
        if self.require_both and not (self.consider_score and self.consider_dice):
            raise ValueError(
                "ThresholdStrategy: require_both=True requires both consider_score=True and consider_dice=True"
            )

    def decide(self, *, turn_score: int, dice_left: int, has_scored: bool, score_needed: int = 0,
               final_round: bool = False, score_to_beat: int = 0, running_total: int = 0) -> bool:
        """Host restatement of the roll/bank rule (strategies.py:212-275) for callers that
        inspect a strategy outside a simulation.  The simulation itself evaluates this on device."""
        del score_needed
        if not has_scored and turn_score < 500:
            return True
        if final_round:
            if running_total <= score_to_beat:
                return True
            if not self.run_up_score:
                return False
        want_s = self.consider_score and turn_score < self.score_threshold
        want_d = self.consider_dice and dice_left > self.dice_threshold
        if self.consider_score and self.consider_dice:
            return (want_s or want_d) if self.require_both else (want_s and want_d)
        if self.consider_score:
            return want_s
        if self.consider_dice:
            return want_d
        return False

    def __str__(self) -> str:
        cs = "S" if self.consider_score else "-"
        cd = "D" if self.consider_dice else "-"
        sf = "F" if self.smart_five else "-"
        so = "O" if self.smart_one else "-"
        rb = "AND" if self.require_both else "OR"
        hd = "H" if self.auto_hot_dice else "-"
        rs = "R" if self.run_up_score else "-"
        fs = "FS" if self.favor_dice_or_score is FavorDiceOrScore.SCORE else "FD"
        return f"Strat({self.score_threshold},{self.dice_threshold})[{cs}{cd}][{sf}{so}{fs}][{rb}][{hd}{rs}]"

    def pack(self, index: int | None = None) -> tuple:
        sid = self.strategy_id if self.strategy_id is not None else (-1 if index is None else index)
        return (int(self.score_threshold), int(self.dice_threshold), int(self.smart_five), int(self.smart_one),
                int(self.consider_score), int(self.consider_dice), int(self.require_both), int(self.auto_hot_dice),
                int(self.run_up_score), int(self.favor_dice_or_score is FavorDiceOrScore.SCORE), int(sid))


@dataclass
class StopAtStrategy(ThresholdStrategy):
    """Named strategy that banks once the turn score crosses a fixed level (strategies.py:293-306)."""

    label: str = ""
    heuristic: bool = False

    def __post_init__(self) -> None:
        super().__post_init__()
        if not re.match(r"stop_at_\d+(?:_heuristic)?\Z", self.label):
            raise ValueError(f"Invalid stop-at strategy label: {self.label!r}")

    def __str__(self) -> str:
        return self.label


_TUPLE_GETTER = None


def strategy_tuple(strategy: ThresholdStrategy) -> tuple:
    global _TUPLE_GETTER
    if _TUPLE_GETTER is None:
        import operator

        _TUPLE_GETTER = operator.attrgetter(*STRATEGY_TUPLE_FIELDS)
    return _TUPLE_GETTER(strategy)


def build_stop_at_strategy(threshold: int, *, heuristic: bool = False,
                           inactive_dice_threshold: int | None = None) -> StopAtStrategy:
    if threshold not in STOP_AT_THRESHOLDS:
        raise ValueError(f"Unregistered stop-at threshold: {threshold}")
    return StopAtStrategy(
        score_threshold=threshold,
        dice_threshold=-1 if inactive_dice_threshold is None else inactive_dice_threshold,
        smart_five=heuristic, smart_one=heuristic, consider_score=True, consider_dice=False, require_both=False,
        auto_hot_dice=heuristic, run_up_score=False, favor_dice_or_score=FavorDiceOrScore.SCORE,
        label=f"stop_at_{threshold}" + ("_heuristic" if heuristic else ""), heuristic=heuristic,
    )


def _coerce_options(options: Sequence[Any] | None, fallback: Iterable[Any]) -> tuple[Any, ...]:
    """strategies.py:314-332 + StrategyGridOptions.from_inputs :556-616: caller lists are sorted so ids
    do not depend on configuration order; tuples and defaults keep their order."""
    if options is None:
        return tuple(fallback)
    values = tuple(options)
    if not isinstance(options, tuple):
        try:
            return tuple(sorted(values))
        except TypeError:
            return values
    return values


def _favor_options(sf: bool, cs: bool, cd: bool) -> tuple[FavorDiceOrScore, ...]:
    if cs and cd:
        return (FavorDiceOrScore.SCORE, FavorDiceOrScore.DICE) if sf else (FavorDiceOrScore.SCORE,)
    if cd and not cs:
        return (FavorDiceOrScore.DICE,)
    return (FavorDiceOrScore.SCORE,)


def iter_strategy_combos(*, score_thresholds, dice_thresholds, smart_five_opts, smart_one_opts, consider_score_opts,
                         consider_dice_opts, auto_hot_dice_opts, run_up_score_opts, inactive_score_threshold: int,
                         inactive_dice_threshold: int, allowed_smart_pairs=None) -> Iterator[tuple]:
    """Deterministic enumeration order; position == strategy_id (strategies.py:346-396)."""
    for sf in smart_five_opts:
        ones = [so for so in smart_one_opts
                if (sf or not so) and (allowed_smart_pairs is None or (sf, so) in allowed_smart_pairs)]
        for so in ones:
            for cs in consider_score_opts:
                score_values = score_thresholds if cs else [inactive_score_threshold]
                for cd in consider_dice_opts:
                    dice_values = dice_thresholds if cd else [inactive_dice_threshold]
                    rb_values = [True, False] if (cs and cd) else [False]
                    favors = _favor_options(sf, cs, cd)
                    for st in score_values:
                        for dt in dice_values:
                            for hd in auto_hot_dice_opts:
                                for rs in run_up_score_opts:
                                    for rb in rb_values:
                                        for ps in favors:
                                            yield (int(st), int(dt), bool(sf), bool(so), bool(cs), bool(cd),
                                                   bool(rb), bool(hd), bool(rs), ps)


def _normalized_options(**kw) -> dict[str, tuple]:
    return {name: _coerce_options(kw.get(name), DEFAULT_STRATEGY_GRID[name]) for name in DEFAULT_STRATEGY_GRID}


def generate_strategy_grid(*, score_thresholds=None, dice_thresholds=None, smart_five_opts=None, smart_one_opts=None,
                           consider_score_opts=(True, False), consider_dice_opts=(True, False),
                           auto_hot_dice_opts=(False, True), run_up_score_opts=(True, False),
                           include_stop_at: bool = False, include_stop_at_heuristic: bool = False):
    """Return ``(strategies, meta)``: the strategy list in id order and a pandas frame describing it
    (simulation.py:55-221)."""
    import pandas as pd

    opts = _normalized_options(score_thresholds=score_thresholds, dice_thresholds=dice_thresholds,
                               smart_five_opts=smart_five_opts, smart_one_opts=smart_one_opts,
                               consider_score_opts=consider_score_opts, consider_dice_opts=consider_dice_opts,
                               auto_hot_dice_opts=auto_hot_dice_opts, run_up_score_opts=run_up_score_opts)
    if not opts["score_thresholds"]:
        raise ValueError("score_thresholds must contain at least one value")
    if not opts["dice_thresholds"]:
        raise ValueError("dice_thresholds must contain at least one value")
    inactive_score = min(opts["score_thresholds"]) - 1
    inactive_dice = min(opts["dice_thresholds"]) - 1
    combos: list[tuple] = list(iter_strategy_combos(inactive_score_threshold=inactive_score,
                                                    inactive_dice_threshold=inactive_dice, **opts))
    extra: list[ThresholdStrategy] = []
    if include_stop_at:
        extra += [build_stop_at_strategy(t, inactive_dice_threshold=inactive_dice) for t in STOP_AT_THRESHOLDS]
    if include_stop_at_heuristic:
        extra += [build_stop_at_strategy(t, heuristic=True, inactive_dice_threshold=inactive_dice)
                  for t in STOP_AT_THRESHOLDS]
    # ids are first-seen positions over grid combos followed by the stop-at tuples (strategies.py:619-712)
    ids: dict[tuple, int] = {}
    for combo in combos + [strategy_tuple(s) for s in extra]:
        ids.setdefault(combo, len(ids))
    # grid combos satisfy the two invariants __post_init__ checks by construction (iter_strategy_combos: smart_one only under
    # smart_five, require_both only when both thresholds are considered), so the 5 160 objects are filled directly: 1 us each
    # instead of the dataclass __init__ + __post_init__'s 8 (40 ms of a `farkle run` on the default grid)
    fields = STRATEGY_TUPLE_FIELDS
    strategies = []
    for combo in combos:
        if (combo[3] and not combo[2]) or (combo[6] and not (combo[4] and combo[5])):
            strategies.append(ThresholdStrategy(*combo, strategy_id=ids[combo]))  # (raises the invariant's ValueError)
            continue
        strat = ThresholdStrategy.__new__(ThresholdStrategy)
        strat.__dict__.update(zip(fields, combo))
        strat.strategy_id = ids[combo]
        strategies.append(strat)
    for s in extra:
        s.strategy_id = ids[strategy_tuple(s)]
        strategies.append(s)
    meta = pd.DataFrame([strategy_tuple(s) for s in strategies], columns=list(STRATEGY_TUPLE_FIELDS))
    meta["strategy_id"] = [s.strategy_id for s in strategies]
    meta["strategy_idx"] = meta.index
    return strategies, meta


def experiment_size(**kw) -> int:
    """Number of strategies a grid configuration yields (simulation.py:224-326)."""
    include = {k: kw.pop(k, False) for k in ("include_stop_at", "include_stop_at_heuristic")}
    pairs = kw.pop("smart_five_and_one_options", None)
    opts = _normalized_options(**kw)
    allowed = None
    if pairs is not None:
        norm = [(bool(a), bool(b)) for a, b in pairs]
        allowed = set(norm)
        opts["smart_five_opts"] = tuple(dict.fromkeys(a for a, _ in norm))
        opts["smart_one_opts"] = tuple(dict.fromkeys(b for _, b in norm))
        if not norm:
            return 0
    n = sum(1 for _ in iter_strategy_combos(inactive_score_threshold=min(opts["score_thresholds"]) - 1,
                                            inactive_dice_threshold=min(opts["dice_thresholds"]) - 1,
                                            allowed_smart_pairs=allowed, **opts))
    return n + len(STOP_AT_THRESHOLDS) * (int(include["include_stop_at"]) + int(include["include_stop_at_heuristic"]))


def default_grid_tuples() -> list[list[int]]:
    """The default 5 160-strategy grid as integer tuples ``[..10 params.., strategy_id]``."""
    opts = _normalized_options()
    combos = iter_strategy_combos(inactive_score_threshold=min(opts["score_thresholds"]) - 1,
                                  inactive_dice_threshold=min(opts["dice_thresholds"]) - 1, **opts)
    return [[c[0], c[1], *(int(v) for v in c[2:9]), int(c[9] is FavorDiceOrScore.SCORE), i]
            for i, c in enumerate(combos)]


def random_threshold_strategy(rng) -> ThresholdStrategy:
    """Sample a strategy from scalar ``rng.integers`` draws in the reference's order (strategies.py:418-452)."""
    sf = bool(rng.integers(0, 2))
    so = bool(rng.integers(0, 2)) if sf else False
    cs = bool(rng.integers(0, 2))
    cd = bool(rng.integers(0, 2))
    rb = bool(rng.integers(0, 2)) if (cs and cd) else False
    if cs == cd:
        fs = FavorDiceOrScore.SCORE if int(rng.integers(0, 2)) == 0 else FavorDiceOrScore.DICE
    else:
        fs = FavorDiceOrScore.SCORE if cs else FavorDiceOrScore.DICE
    return ThresholdStrategy(score_threshold=int(rng.integers(1, 20)) * 50, dice_threshold=int(rng.integers(0, 5)),
                             smart_five=sf, smart_one=so, consider_score=cs, consider_dice=cd, require_both=rb,
                             favor_dice_or_score=fs)


def prepare_public_helper_strategies(strategies: Sequence[ThresholdStrategy]) -> list[ThresholdStrategy]:
    """Copies with unique integer ids; missing ids get the smallest unused non-negative integers in input
    order (simulation.py:361-409)."""
    used: set[int] = set()
    provided: list[int | None] = []
    for pos, s in enumerate(strategies):
        if s.strategy_id is None:
            provided.append(None)
            continue
        sid = s.strategy_id
        if isinstance(sid, bool) or not isinstance(sid, (int, np.integer)) or int(sid) < 0 or int(sid) > 2**31 - 1:
            raise ValueError(f"strategies[{pos}].strategy_id must be a canonical non-negative int32")
        if int(sid) in used:
            raise ValueError(f"Caller-provided strategy IDs must be unique; found {int(sid)}")
        used.add(int(sid))
        provided.append(int(sid))
    out, nxt = [], 0
    for s, sid in zip(strategies, provided):
        if sid is None:
            while nxt in used:
                nxt += 1
            sid = nxt
            used.add(sid)
            nxt += 1
        # (dataclasses.replace re-runs __init__ + __post_init__ on values that were validated when `s` was made)
        if type(s) is ThresholdStrategy:
            twin = ThresholdStrategy.__new__(ThresholdStrategy)
            twin.__dict__.update(s.__dict__)
        else:
            twin = _copy.copy(s)
        twin.strategy_id = sid
        out.append(twin)
    return out


def pack_strategies(strategies: Sequence[ThresholdStrategy]) -> np.ndarray:
    """Struct-of-records array (``fk_strategy[S]``) handed to the C-ABI."""
    out = np.zeros(len(strategies), dtype=STRATEGY_DTYPE)
    for i, s in enumerate(strategies):
        out[i] = s.pack(i)
    return out
