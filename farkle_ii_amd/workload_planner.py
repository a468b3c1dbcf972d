"""Deterministic workload planning for broad tournament screening.

Mirrors ``src/farkle/simulation/workload_planner.py`` (:19-216): Wilson-width target -> number of shuffles ->
equal contiguous batches.  Host-only arithmetic (scipy for the normal quantile, as in the reference).
"""
from __future__ import annotations

import json
import functools
import math
import os
from dataclasses import asdict, dataclass, replace
from pathlib import Path

WORKLOAD_PLAN_VERSION = 1
CAP_CONFIG_KEY = "screening.max_shuffles_per_root_k"


@dataclass(frozen=True)
class TournamentWorkloadPlan:
    root_seed: int
    k: int
    strategy_count: int
    confidence: float
    resolution_delta: float
    required_shuffles_unrounded: int
    required_shuffles: int
    batch_count: int
    shuffles_per_batch: int
    batch_construction: str
    games_per_shuffle: int
    required_games: int
    achieved_resolution: float
    shuffle_cap: int | None
    cap_exceeded: bool
    achieved_resolution_at_cap: float | None
    projected_games_per_second: float | None = None
    projected_runtime_seconds: float | None = None
    plan_version: int = WORKLOAD_PLAN_VERSION

    @property
    def status(self) -> str:
        return "blocked_by_cap" if self.cap_exceeded else "not_started"

    def with_games_per_second(self, games_per_second: float) -> "TournamentWorkloadPlan":
        if not math.isfinite(games_per_second) or games_per_second <= 0.0:
            raise ValueError("games_per_second must be finite and positive")
        return replace(self, projected_games_per_second=float(games_per_second),
                       projected_runtime_seconds=self.required_games / float(games_per_second))

    def to_dict(self) -> dict[str, object]:
        return {**asdict(self), "status": self.status, "cap_config_key": CAP_CONFIG_KEY}


class WorkloadCapExceeded(RuntimeError):
    def __init__(self, plan: TournamentWorkloadPlan) -> None:
        self.plan = plan
        super().__init__(f"Required {plan.required_shuffles} shuffles for root={plan.root_seed}, k={plan.k}, but "
                         f"{CAP_CONFIG_KEY}={plan.shuffle_cap}. Raise {CAP_CONFIG_KEY} to at least "
                         f"{plan.required_shuffles} and resume.")


def _is_int(value, minimum: int) -> bool:
    return isinstance(value, int) and not isinstance(value, bool) and value >= minimum


def _check_level(confidence: float) -> None:
    if not 0.0 < confidence < 1.0:
        raise ValueError("confidence must be between 0 and 1")


@functools.lru_cache(maxsize=64)
def _normal_quantile(q: float) -> float:
    """scipy.stats.norm.ppf(q) — the reference's call (workload_planner.py:86), evaluated once per level: the resolution search asks
    for the same quantile at every candidate size (39 calls of 0.1 ms each per plan)."""
    from scipy.stats import norm

    return float(norm.ppf(q))


def worst_case_wilson_width(n: int, *, confidence: float = 0.95) -> float:
    """Widest full Wilson score interval any success count can produce at sample size ``n``: the interval is widest at
    p = 1/2, i.e. at floor(n/2) or ceil(n/2) successes (workload_planner.py:88-108)."""
    if not _is_int(n, 1):
        raise ValueError("n must be a positive integer")
    _check_level(confidence)
    z = _normal_quantile(0.5 + confidence / 2.0)
    zz = z * z
    # The value lands in simulation_workload_plan.json (achieved_resolution) and decides minimum_shuffles_for_resolution, so
    # every operation is in the reference's order (workload_planner.py:87-93): the Wilson radicand is
    # p(1-p)/n + z^2/(4 n^2) with the product 4.0 * n * n formed first — z^2/n/(4n) rounds differently at 1 in ~9 000 sizes.
    widths = []
    for successes in (n // 2, (n + 1) // 2):
        p_hat = successes / n
        shrink = 1.0 + zz / n
        half = z * math.sqrt(p_hat * (1.0 - p_hat) / n + zz / (4.0 * n * n))
        widths.append(2.0 * half / shrink)
    return max(widths)


def minimum_shuffles_for_resolution(resolution_delta: float, *, confidence: float = 0.95) -> int:
    """Smallest n whose worst-case Wilson width is <= ``resolution_delta`` (the width is decreasing in n)."""
    if not 0.0 < resolution_delta < 1.0:
        raise ValueError("resolution_delta must be between 0 and 1")
    _check_level(confidence)

    def fine_enough(n: int) -> bool:
        return worst_case_wilson_width(n, confidence=confidence) <= resolution_delta

    hi = 1
    while not fine_enough(hi):  # gallop to an upper bracket, then bisect (lo fails, hi passes)
        hi *= 2
    lo = hi // 2
    while hi - lo > 1:
        mid = (lo + hi) >> 1
        lo, hi = (lo, mid) if fine_enough(mid) else (mid, hi)
    return hi


def plan_tournament_workload(*, root_seed: int, k: int, strategy_count: int, resolution_delta: float, confidence: float = 0.95,
                             batch_count: int = 100, min_shuffles_per_batch: int = 30, shuffle_cap: int | None = None,
                             projected_games_per_second: float | None = None) -> TournamentWorkloadPlan:
    """Shuffles per (root, k) from the screening resolution, rounded up to ``batch_count`` equal contiguous batches
    (workload_planner.py:130-193)."""
    if not _is_int(k, 2):
        raise ValueError("k must be an integer of at least 2")
    if not _is_int(strategy_count, k) or strategy_count % k:
        raise ValueError("strategy_count must be a positive multiple of k")
    if not _is_int(batch_count, 2):
        raise ValueError("batch_count must be an integer of at least 2")
    if not _is_int(min_shuffles_per_batch, 1):
        raise ValueError("min_shuffles_per_batch must be a positive integer")
    if shuffle_cap is not None and not _is_int(shuffle_cap, 1):
        raise ValueError("shuffle_cap must be positive when configured")
    needed = minimum_shuffles_for_resolution(resolution_delta, confidence=confidence)
    per_batch = max(min_shuffles_per_batch, -(-needed // batch_count))
    total = per_batch * batch_count
    games_per_shuffle = strategy_count // k
    over_cap = shuffle_cap is not None and total > shuffle_cap
    fields = dict(root_seed=int(root_seed), k=k, strategy_count=strategy_count, confidence=float(confidence),
                  resolution_delta=float(resolution_delta), required_shuffles_unrounded=needed, required_shuffles=total,
                  batch_count=batch_count, shuffles_per_batch=per_batch, batch_construction="equal_contiguous",
                  games_per_shuffle=games_per_shuffle, required_games=total * games_per_shuffle,
                  achieved_resolution=worst_case_wilson_width(total, confidence=confidence), shuffle_cap=shuffle_cap,
                  cap_exceeded=over_cap,
                  achieved_resolution_at_cap=worst_case_wilson_width(shuffle_cap, confidence=confidence) if over_cap else None)
    plan = TournamentWorkloadPlan(**fields)
    return plan if projected_games_per_second is None else plan.with_games_per_second(projected_games_per_second)


def workload_plan_bytes(plan: TournamentWorkloadPlan) -> bytes:
    """The plan file's exact bytes: canonical JSON, indent 2, sorted keys, trailing newline (simulation/runner.py:408)."""
    return json.dumps(plan.to_dict(), indent=2, sort_keys=True).encode("utf-8") + b"\n"


def write_workload_plan(path: Path, plan: TournamentWorkloadPlan) -> None:
    """Atomic write of ``workload_plan_bytes``."""
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    tmp = path.with_name(path.name + ".tmp")
    tmp.write_bytes(workload_plan_bytes(plan))
    os.replace(tmp, path)
