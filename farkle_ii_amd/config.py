"""Application configuration for the simulation path.

Mirrors the sections of ``src/farkle/config.py`` this path reads — ``io`` :146-150, ``sim`` :230-289, ``screening``
:172-184, ``batching`` :188-192, ``rng`` :154-158 — with the same YAML overlay semantics (``load_app_config`` :1494,
dotted keys, later overlays win), ``--set section.key=value`` coercion (``apply_dot_overrides`` :1692) and path helpers
(``results_root`` :484-492, ``n_dir`` :858, ``simulation_row_dir`` :862-879, ``checkpoint_path`` :881).  Sections that
only configure the reference's analysis pipeline are accepted and carried opaquely.  Superset: ``--set`` also accepts
list values (``sim.n_players_list=[2]``), which the reference can only take from YAML.
"""
from __future__ import annotations

import ast
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import Any, Mapping, Sequence

import yaml

_OPAQUE_SECTIONS = {"analysis", "ingest", "combine", "trueskill", "head2head", "hgb", "orchestration", "resources", "profile",
                    "robustness", "artifact_contract", "k_aggregation"}


@dataclass
class IOConfig:
    results_dir_prefix: Path = Path("results")
    analysis_subdir: str = "analysis"


@dataclass
class RNGConfig:
    scheme_version: int = 2
    bit_generator: str = "PCG64DXSM"


@dataclass
class ScreeningConfig:
    resolution_delta: float = 0.03
    interval_confidence: float = 0.95
    practical_delta_by_k: dict | None = None
    delta_across_k: float | None = 0.03
    bootstrap_replicates: int = 2_000
    candidate_contribution_size: int = 75
    controls: list = field(default_factory=list)
    mandatory_diagnostics: list = field(default_factory=list)
    max_shuffles_per_root_k: int | None = None
    projected_games_per_second: float | None = None


@dataclass
class BatchingConfig:
    target_batches: int = 100
    min_shuffles_per_batch: int = 30


@dataclass
class SimConfig:
    n_players_list: list[int] = field(default_factory=lambda: [5])
    seed: int = 0
    seed_list: list[int] | None = None
    expanded_metrics: bool = False
    row_dir: Path | None = None
    metric_chunk_dir: Path | None = None
    sidecars: bool = False  # this engine's option: per-artifact <name>.sidecar.json (sidecars.py; contract version 2)
    all_player_batch_dir: Path | None = None  # this engine's option: all-player batch metrics without rows (all_player.py)
    rng_lag_sums: bool = False  # this engine's option: lag sufficient statistics of the RNG diagnostics' strategy family (rng_lags.py)
    per_n: dict = field(default_factory=dict)
    n_jobs: int | None = None
    mp_start_method: str | None = None
    desired_sec_per_chunk: int = 10
    ckpt_every_sec: int = 30
    progress_logging: dict = field(default_factory=dict)
    score_thresholds: list[int] | None = None
    dice_thresholds: list[int] | None = None
    smart_five_opts: Sequence[bool] | None = None
    smart_one_opts: Sequence[bool] | None = None
    consider_score_opts: Sequence[bool] = (True, False)
    consider_dice_opts: Sequence[bool] = (True, False)
    auto_hot_dice_opts: Sequence[bool] = (True, False)
    run_up_score_opts: Sequence[bool] = (True, False)
    include_stop_at: bool = False
    include_stop_at_heuristic: bool = False

    def resolve_seed_list(self, expected_len: int) -> list[int]:
        if expected_len < 1:
            raise ValueError("expected_len must be >= 1")
        if self.seed_list is not None:
            if len(self.seed_list) != expected_len:
                raise ValueError(f"sim.seed_list must contain exactly {expected_len} seeds, got {self.seed_list!r}")
            return list(self.seed_list)
        if expected_len == 1:
            return [self.seed]
        raise ValueError(f"sim.seed_list must be set for orchestration requiring {expected_len} seeds")

    def populate_seed_list(self, expected_len: int) -> list[int]:
        seeds = self.resolve_seed_list(expected_len)
        self.seed_list = list(seeds)
        if expected_len in {1, 2}:
            self.seed = seeds[0]
        return seeds


@dataclass
class AppConfig:
    io: IOConfig = field(default_factory=IOConfig)
    sim: SimConfig = field(default_factory=SimConfig)
    rng: RNGConfig = field(default_factory=RNGConfig)
    screening: ScreeningConfig = field(default_factory=ScreeningConfig)
    batching: BatchingConfig = field(default_factory=BatchingConfig)
    opaque: dict = field(default_factory=dict)  # analysis-only sections, untouched
    # run identities a caller hands in (config.py:472-474 of the reference: private, never part of a configuration digest) — what
    # artifact-contract version 3 signs with (contract_v3.py)
    _code_identity: dict | None = field(default=None, init=False, repr=False, compare=False)
    _run_lineage_sha256: str | None = field(default=None, init=False, repr=False, compare=False)
    _game_profile_sha256: str | None = field(default=None, init=False, repr=False, compare=False)

    @property
    def results_root(self) -> Path:
        base = Path(self.io.results_dir_prefix)
        if not base.is_absolute():
            base = Path("data") / base
        suffix = f"_seed_{self.sim.seed}"
        if base.name.endswith(suffix):
            return base
        return base.parent / f"{base.name}{suffix}"

    def n_dir(self, n: int) -> Path:
        return self.results_root / f"{n}_players"

    @property
    def artifact_contract_version(self) -> int:
        """The contract the sidecar writers follow.  ``artifact_contract.artifact_contract_version`` when the configuration states it
        (the section is carried opaquely; the reference's default is 3, config.py:210); otherwise 3 when the caller supplied the code
        identity version 3 signs with, and the structural version 2 when not."""
        stated = (self.opaque.get("artifact_contract") or {}).get("artifact_contract_version")
        if stated is not None:
            return int(stated)
        return 3 if self._code_identity is not None else 2

    def _per_n_dir(self, raw_value, n: int, what: str) -> Path | None:
        if not raw_value:
            return None
        raw_text = str(raw_value)
        try:
            formatted = raw_text.format(n=n, n_players=n, p=f"{n}p")
        except KeyError as exc:
            raise ValueError(f"unknown simulation {what} placeholder: {exc.args[0]}") from exc
        path = Path(formatted)
        if formatted == raw_text and path.name and not path.name.startswith(f"{n}p"):
            path = path.parent / f"{n}p_{path.name}"
        return path if path.is_absolute() else self.n_dir(n) / path

    def simulation_row_dir(self, n: int) -> Path | None:
        return self._per_n_dir(self.sim.row_dir, n, "row-dir")

    def metric_chunk_dir(self, n: int) -> Path | None:
        return self._per_n_dir(self.sim.metric_chunk_dir, n, "metric-chunk-dir")

    def all_player_batch_dir(self, n: int) -> Path | None:
        return self._per_n_dir(self.sim.all_player_batch_dir, n, "all-player-batch-dir")

    def rng_diagnostic_lags(self) -> tuple[int, ...]:
        """``analysis.rng_diagnostic_lags`` (config.py:335, validated like :1933-1939); the analysis section is carried opaquely."""
        raw = (self.opaque.get("analysis") or {}).get("rng_diagnostic_lags", (1,))
        lags = tuple(int(v) for v in raw)
        if not lags or any(v < 1 for v in lags) or tuple(sorted(set(lags))) != lags:
            raise ValueError("analysis.rng_diagnostic_lags must be unique increasing positive integers")
        return lags

    def rng_lag_sums_path(self, n: int) -> Path:
        return self.n_dir(n) / f"{n}p_rng_lag_sums.parquet"

    def rng_lag_stats_path(self, n: int) -> Path:
        return self.n_dir(n) / f"{n}p_rng_lag_stats.parquet"

    def checkpoint_path(self, n: int) -> Path:
        return self.n_dir(n) / f"{n}p_checkpoint.pkl"

    def metrics_path(self, n: int) -> Path:
        return self.n_dir(n) / f"{n}p_metrics.parquet"

    def strategy_manifest_root_path(self) -> Path:
        return self.results_root / "strategy_manifest.parquet"


def expand_dotted_keys(mapping: Mapping[str, Any]) -> dict[str, Any]:
    out: dict[str, Any] = {}
    for key, value in mapping.items():
        if isinstance(value, Mapping):
            value = expand_dotted_keys(value)
        parts = str(key).split(".")
        cur = out
        for part in parts[:-1]:
            cur = cur.setdefault(part, {})
            if not isinstance(cur, dict):
                raise ValueError(f"dotted key {key!r} collides with a scalar")
        if isinstance(value, dict) and isinstance(cur.get(parts[-1]), dict):
            cur[parts[-1]] = _deep_merge(cur[parts[-1]], value)
        else:
            cur[parts[-1]] = value
    return out


def _deep_merge(base: dict, overlay: Mapping[str, Any]) -> dict:
    out = dict(base)
    for key, value in overlay.items():
        if isinstance(value, Mapping) and isinstance(out.get(key), dict):
            out[key] = _deep_merge(out[key], value)
        else:
            out[key] = value
    return out


def _fill(section_obj, data: Mapping[str, Any], section_name: str) -> None:
    known = {f.name for f in fields(section_obj)}
    for key, value in data.items():
        if key not in known:
            raise ValueError(f"Unknown option {key!r} in config section {section_name!r}")
        if key in {"row_dir", "metric_chunk_dir", "all_player_batch_dir", "results_dir_prefix"} and value is not None:
            value = Path(value)
        setattr(section_obj, key, value)


def load_app_config(*overlays: Path, seed_list_len: int | None = None) -> AppConfig:
    """Merge YAML overlays (later wins) into an :class:`AppConfig`."""
    data: dict[str, Any] = {}
    for path in overlays:
        with Path(path).open("r", encoding="utf-8") as fh:
            # libyaml's loader when the wheel has it: the same document model, a tenth of the time (4 ms of a 30-ms `farkle run`)
            overlay = yaml.load(fh, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader)) or {}
        if not isinstance(overlay, Mapping):
            raise TypeError(f"Config file {path} must contain a mapping")
        data = _deep_merge(data, expand_dotted_keys(overlay))
    cfg = AppConfig()
    for name, section in data.items():
        if name in _OPAQUE_SECTIONS:
            cfg.opaque[name] = section
            continue
        if name not in {"io", "sim", "rng", "screening", "batching"}:
            raise ValueError(f"Unknown config section {name!r}")
        if not isinstance(section, Mapping):
            raise TypeError(f"Config section {name!r} must be a mapping")
        _fill(getattr(cfg, name), section, name)
    players = cfg.sim.n_players_list
    if isinstance(players, list):
        norm = []
        for entry in players:
            try:
                value = int(entry)
            except (TypeError, ValueError) as exc:
                raise ValueError(f"invalid n_players_list entry: {entry!r}") from exc
            if value < 2:
                raise ValueError("sim.n_players_list requires concrete player counts >= 2; select cross-k work with "
                                 "canonical scope settings")
            norm.append(value)
        cfg.sim.n_players_list = norm
    if cfg.sim.seed_list is not None:  # sim.seed_list is canonical; seed = seed_list[0] (config.py:1455-1491)
        cfg.sim.seed_list = [int(s) for s in cfg.sim.seed_list]
        if seed_list_len is not None and len(cfg.sim.seed_list) != seed_list_len:
            raise ValueError(f"load_app_config: sim.seed_list must contain exactly {seed_list_len} seeds, "
                             f"got {cfg.sim.seed_list!r}")
        if cfg.sim.seed_list:
            cfg.sim.seed = cfg.sim.seed_list[0]
    if seed_list_len is not None:
        cfg.sim.populate_seed_list(seed_list_len)
    if cfg.rng.scheme_version != 2 or cfg.rng.bit_generator != "PCG64DXSM":
        raise ValueError("this engine implements RNG scheme v2 with PCG64DXSM only")
    return cfg


def _coerce(raw: str, current: Any) -> Any:
    text = raw.strip()
    if text.startswith("[") or text.startswith("{") or text.startswith("("):
        return ast.literal_eval(text)  # superset of the reference: list-valued overrides
    if isinstance(current, bool):
        low = text.lower()
        if low in {"1", "true", "yes", "on"}:
            return True
        if low in {"0", "false", "no", "off"}:
            return False
        raise ValueError(f"Cannot parse boolean value from {raw!r}")
    if isinstance(current, int):
        return int(text)
    if isinstance(current, float):
        return float(text)
    if isinstance(current, Path):
        return Path(text)
    if current is None:
        if text.lower() in {"null", "none"}:
            return None
        for cast in (int, float):
            try:
                return cast(text)
            except ValueError:
                pass
    return text


def apply_dot_overrides(cfg: AppConfig, pairs: Sequence[str]) -> AppConfig:
    """Apply ``section.option=value`` overrides to ``cfg``."""
    for pair in pairs:
        if "=" not in pair:
            raise ValueError(f"Invalid override {pair!r}")
        key, raw = pair.split("=", 1)
        if "." not in key:
            raise ValueError(f"Invalid override {pair!r}")
        section_name, option = key.split(".", 1)
        if section_name in _OPAQUE_SECTIONS:
            cfg.opaque.setdefault(section_name, {})[option] = raw
            continue
        section = getattr(cfg, section_name, None)
        if section is None or section_name == "opaque":
            raise AttributeError(f"Unknown config section {section_name!r}")
        if not hasattr(section, option):
            raise AttributeError(f"Unknown option {option!r} in section {section_name!r}")
        value = _coerce(raw, getattr(section, option))
        if option in {"row_dir", "metric_chunk_dir", "all_player_batch_dir", "results_dir_prefix"} and value is not None:
            value = Path(value)
        setattr(section, option, value)
    return cfg
