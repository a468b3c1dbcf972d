"""Tournament checkpoints that both engines can read.

The reference persists ``{k}p_checkpoint.pkl`` as a pickle of ``{"win_totals", "outcome_counts", "metric_sums",
"metric_square_sums", "meta"}`` (``src/farkle/simulation/run_tournament.py:622-651``) and, on resume, normalises whatever
it finds through ``_coerce_counter`` / ``_coerce_metric_sums`` (``:654-766``): ``win_totals`` may be any ``Counter`` or
mapping as long as ``outcome_counts`` carries the exposure counters.  This module

* writes exactly that payload with PLAIN containers (``collections.Counter`` / ``dict``), so the file unpickles in a
  process that has neither this package nor the reference installed, and the reference resumes from it;
* reads the reference's own checkpoints: its ``OutcomeCounter`` pickles through
  ``farkle.simulation.run_tournament._restore_outcome_counter``, which a restricted unpickler maps onto this package's
  counter type (no code of the pickle's choosing is ever imported);
* converts between the payload and the engine's ``int64 [S][26]`` tally (every stored value is an integer, the reference
  keeps the metric sums as exact float64).
"""
from __future__ import annotations

import io
import pickle
from collections import Counter, defaultdict
from pathlib import Path
from typing import Any, Mapping, Sequence

import numpy as np

from .backend import COL_ATTEMPTED, COL_COMPLETED, COL_SAFETY, COL_SQ_SUMS, COL_SUMS, COL_WINS

_SAFE_BUILTINS = {
    ("collections", "Counter"): Counter, ("collections", "defaultdict"): defaultdict, ("collections", "OrderedDict"): dict,
    ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "set"): set, ("builtins", "frozenset"): frozenset,
    ("builtins", "tuple"): tuple, ("builtins", "int"): int, ("builtins", "float"): float, ("builtins", "str"): str,
    ("builtins", "bool"): bool, ("builtins", "bytes"): bytes, ("builtins", "complex"): complex,
}
# where an OutcomeCounter may have been pickled from: the reference, and this package's earlier releases
_COUNTER_MODULES = ("farkle.simulation.run_tournament", "farkle_ii_amd.tournament")


class _CheckpointUnpickler(pickle.Unpickler):
    """Containers and numbers only; the two OutcomeCounter entry points resolve to this package's implementation."""

    def find_class(self, module: str, name: str):
        if module in _COUNTER_MODULES and name in ("_restore_outcome_counter", "OutcomeCounter"):
            from . import tournament

            return getattr(tournament, name)
        if (module, name) in _SAFE_BUILTINS:
            return _SAFE_BUILTINS[(module, name)]
        if module == "numpy.core.multiarray" or module == "numpy._core.multiarray":
            if name in ("_reconstruct", "scalar"):
                import importlib

                return getattr(importlib.import_module(module), name)
        if module == "numpy" and name in ("ndarray", "dtype"):
            return getattr(np, name)
        raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}, which a tournament checkpoint never contains")


def load_checkpoint(path: Path) -> dict[str, Any]:
    """Unpickle a checkpoint written by this engine or by the reference."""
    payload = _CheckpointUnpickler(io.BytesIO(Path(path).read_bytes())).load()
    if not isinstance(payload, Mapping):
        raise TypeError(f"{path} does not hold a checkpoint payload")
    return dict(payload)


def _key(k: Any) -> Any:
    return int(k) if str(k).lstrip("-").isdigit() else k


def payload_to_tally(payload: Mapping[str, Any], ids: Sequence[int], labels: Sequence[str]) -> np.ndarray:
    """Checkpoint payload -> ``int64 [S][26]`` (wins, attempted, completed, safety, sums, square sums), validated the
    way ``_coerce_counter`` validates it (run_tournament.py:701-744)."""
    index = {int(s): i for i, s in enumerate(ids)}
    tally = np.zeros((len(ids), 26), dtype=np.int64)

    def put(values: Mapping[Any, Any] | None, col: int, what: str) -> None:
        for key, v in (values or {}).items():
            key = _key(key)
            if key not in index:
                raise ValueError(f"checkpoint {what} names strategy {key!r}, which the configured grid does not contain")
            f = float(v)
            if f != int(f) or f < 0:
                raise ValueError(f"checkpoint {what}[{key}] = {v!r} is not a non-negative integer")
            tally[index[key], col] = int(f)

    raw = payload.get("win_totals", {})
    put(dict(raw), COL_WINS, "win_totals")
    oc = payload.get("outcome_counts")
    if oc is None and all(hasattr(raw, a) for a in ("attempted_exposures", "completed_exposures", "safety_limit_exposures")):
        oc = {"attempted_exposures": dict(raw.attempted_exposures), "completed_exposures": dict(raw.completed_exposures),
              "safety_limit_exposures": dict(raw.safety_limit_exposures)}
    if oc is None:
        raise ValueError("checkpoint has no outcome_counts: exposures cannot be restored")
    put(oc.get("attempted_exposures"), COL_ATTEMPTED, "attempted_exposures")
    put(oc.get("completed_exposures"), COL_COMPLETED, "completed_exposures")
    put(oc.get("safety_limit_exposures"), COL_SAFETY, "safety_limit_exposures")
    sums = payload.get("metric_sums") or {}
    sqs = payload.get("metric_square_sums") or payload.get("metric_sq_sums") or {}
    for j, label in enumerate(labels):
        put(sums.get(label), COL_SUMS + j, f"metric_sums[{label}]")
        put(sqs.get(label), COL_SQ_SUMS + j, f"metric_square_sums[{label}]")
    if not np.array_equal(tally[:, COL_ATTEMPTED], tally[:, COL_COMPLETED] + tally[:, COL_SAFETY]) or (tally[:, COL_WINS] > tally[:, COL_COMPLETED]).any():
        raise ValueError("checkpoint strategy exposure conservation failed")
    return tally


def dump_checkpoint(wins, sums: Mapping[str, Mapping[Any, float]] | None, sq_sums: Mapping[str, Mapping[Any, float]] | None,
                    meta: Mapping[str, Any]) -> bytes:
    """The reference's payload (run_tournament.py:622-651) in plain containers: ``win_totals`` a ``collections.Counter``,
    the exposure counters in ``outcome_counts`` (what ``_coerce_counter`` restores them from)."""
    payload: dict[str, Any] = {"win_totals": Counter({k: int(v) for k, v in wins.items()}), "outcome_counts": wins.outcome_payload()}
    if sums is not None and sq_sums is not None:
        payload["metric_sums"] = {m: {k: float(v) for k, v in d.items()} for m, d in sums.items()}
        payload["metric_square_sums"] = {m: {k: float(v) for k, v in d.items()} for m, d in sq_sums.items()}
    payload["meta"] = dict(meta)
    return pickle.dumps(payload, protocol=pickle.HIGHEST_PROTOCOL)
