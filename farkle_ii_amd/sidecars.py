"""Per-artifact sidecars of the simulation outputs (SURVEY section 8, f3 — the part that needs no release identity).

The reference publishes every simulation artifact with an adjacent ``<name>.sidecar.json`` built by
``_simulation_output_sidecar`` (``src/farkle/simulation/runner.py:338-376`` -> ``make_artifact_sidecar``,
``src/farkle/utils/artifact_contract.py:288-359``): the producer / operation / estimand metadata of the artifact plus its
byte identity (SHA-256 and size), serialised as canonical JSON (``_canonical_json`` :495-499) and checked by
``validate_artifact_sidecar`` (:629-668).  This module writes that JSON for the artifacts of this engine's runs, at the
reference's contract version 2 (the structural contract of ``_validate_sidecar_fields`` :388-492), with ``code_revision`` =
this engine's identity.

NOT built: contract version 3 (``utils/release_identity.py`` / ``utils/authenticated_contract.py``) and the authenticated
``simulation.done.json`` stamp of ``write_stage_done`` — both sign with the reference's own Git checkout identity, which an
independent engine cannot and should not forge.
"""
from __future__ import annotations

import hashlib
import json
import os
from pathlib import Path
from typing import Any, Mapping, Sequence

from . import __version__
from .rows import OUTCOME_SCHEMA_VERSION, TOURNAMENT_METHOD_VERSION

SIDECAR_SUFFIX = ".sidecar.json"
ARTIFACT_CONTRACT_VERSION = 2  # artifact_contract.py:39
ESTIMAND_VERSION = 2           # ArtifactContractConfig defaults (config.py)
SCHEMA_VERSION = 2
RNG_SCHEME_VERSION = 2
SIDECAR_FIELDS = (
    "artifact_contract_version", "estimand_version", "schema_version", "artifact_name", "producer", "scope", "source_scope", "operation",
    "method_contract", "baseline", "weighted_quantity", "k_aggregation_method", "k_weights", "support_count_role", "uncertainty_method",
    "replication_unit", "conditioning", "consistency_columns", "source_artifacts", "grouping_keys", "player_counts", "required_player_counts",
    "missing_cell_policy", "seed_scope", "rng_scheme_version", "config_hash", "input_manifest_hashes", "code_revision", "artifact_sha256",
    "artifact_size_bytes")
OPERATIONS = {  # artifact kind -> the reference's operation identifier (runner.py:424, 624, 1438, 1477, 1488, 1499, 1649, 1708)
    "strategy_manifest": "publish_strategy_manifest", "workload_plan": "publish_simulation_workload_plan",
    "checkpoint": "publish_simulation_checkpoint", "row_shard": "publish_simulation_row_shard",
    "metric_chunk": "publish_simulation_metric_chunk", "shard_manifest": "publish_simulation_shard_manifest",
    "checkpoint_summary": "publish_simulation_checkpoint_summary", "metrics_summary": "publish_simulation_metrics_summary"}


def sidecar_path(artifact_path: Path | str) -> Path:
    path = Path(artifact_path)
    return path.with_name(f"{path.name}{SIDECAR_SUFFIX}")


def sha256_file(path: Path | str) -> str:
    digest = hashlib.sha256()
    with open(path, "rb") as fh:
        while chunk := fh.read(1 << 20):
            digest.update(chunk)
    return digest.hexdigest()


def engine_code_revision() -> str:
    """This engine's identity: package version + SHA-256 of the device sources the artifact was computed by."""
    h = hashlib.sha256()
    src = Path(__file__).resolve().parent / "csrc"
    for name in ("fk_kernels.h", "fk_play_hc.h", "fk_device.h", "farkle_hip.hip"):
        if (src / name).exists():
            h.update((src / name).read_bytes())
    return f"farkle_ii_amd-{__version__}+kernels.{h.hexdigest()[:16]}"


def config_hash(cfg) -> str:
    """SHA-256 over this engine's effective configuration (compact canonical JSON, as compute_config_sha does for the
    reference's AppConfig, config.py:2094-2104; the two configurations have different fields, so the digests differ)."""
    import dataclasses

    def plain(obj: Any) -> Any:
        if dataclasses.is_dataclass(obj):
            return {f.name: plain(getattr(obj, f.name)) for f in dataclasses.fields(obj) if not f.name.startswith("_")}
        if isinstance(obj, Mapping):
            return {str(k): plain(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [plain(v) for v in obj]
        if isinstance(obj, Path):
            return str(obj)
        return obj

    return hashlib.sha256(json.dumps(plain(cfg), sort_keys=True, separators=(",", ":"), ensure_ascii=False, default=str).encode("utf-8")).hexdigest()


def simulation_output_sidecar(cfg, path: Path | str, *, n_players: int, operation: str, sources: Sequence[Path | str] = (),
                              support_counts: Sequence[int] | None = None) -> dict[str, Any]:
    """The producer contract of one simulation artifact, field for field what ``_simulation_output_sidecar`` builds
    (runner.py:338-376); the artifact identity is bound by :func:`bind_artifact`."""
    counts = sorted({int(c) for c in (support_counts if support_counts is not None else [n_players])})
    return {
        "artifact_contract_version": ARTIFACT_CONTRACT_VERSION, "estimand_version": ESTIMAND_VERSION, "schema_version": SCHEMA_VERSION,
        "artifact_name": Path(path).name, "producer": "simulation", "scope": "diagnostics", "source_scope": "diagnostics",
        "operation": operation,
        "method_contract": {"kind": "operation", "procedure": operation,
                            "parameters": {"tournament_method_version": TOURNAMENT_METHOD_VERSION,
                                           "outcome_schema_version": OUTCOME_SCHEMA_VERSION}},
        "baseline": "tournament_design", "weighted_quantity": "raw_simulation_evidence", "k_aggregation_method": "none", "k_weights": None,
        "support_count_role": "root_k_simulation", "uncertainty_method": "deterministic_monte_carlo", "replication_unit": "shuffle",
        "conditioning": "all_attempted_games", "consistency_columns": [], "source_artifacts": [str(Path(p)) for p in sources],
        "grouping_keys": [], "player_counts": counts, "required_player_counts": counts, "missing_cell_policy": "fail",
        "seed_scope": "single_root", "rng_scheme_version": RNG_SCHEME_VERSION, "config_hash": config_hash(cfg), "input_manifest_hashes": [],
        "code_revision": engine_code_revision(), "artifact_sha256": "", "artifact_size_bytes": 0}


def bind_artifact(sidecar: Mapping[str, Any], path: Path | str) -> dict[str, Any]:
    """``ArtifactSidecar.with_artifact_identity`` (artifact_contract.py:244-252): name, SHA-256 and size of the bytes at ``path``."""
    path = Path(path)
    return {**sidecar, "artifact_name": path.name, "artifact_sha256": sha256_file(path), "artifact_size_bytes": path.stat().st_size}


def canonical_json(sidecar: Mapping[str, Any]) -> str:
    return json.dumps({k: sidecar[k] for k in SIDECAR_FIELDS}, indent=2, sort_keys=True, ensure_ascii=False) + "\n"


def write_sidecar(path: Path | str, template: Mapping[str, Any]) -> Path:
    """Bind ``template`` to the bytes at ``path`` and write ``<path>.sidecar.json`` atomically; returns the sidecar path."""
    out = sidecar_path(path)
    tmp = out.with_name(out.name + ".tmp")
    tmp.write_text(canonical_json(bind_artifact(template, path)), encoding="utf-8")
    os.replace(tmp, out)
    return out


def validate_sidecar(path: Path | str, expected: Mapping[str, Any] | None = None) -> dict[str, Any]:
    """The checks of ``validate_artifact_sidecar`` (artifact_contract.py:629-668) that matter for a consumer: the sidecar is
    there, complete, of contract version 2, names this artifact and matches its bytes; ``expected`` fields are equal."""
    path = Path(path)
    try:
        payload = json.loads(sidecar_path(path).read_text(encoding="utf-8"))
    except FileNotFoundError as exc:
        raise ValueError(f"missing sidecar for {path}; expected adjacent {sidecar_path(path).name}") from exc
    if set(payload) != set(SIDECAR_FIELDS):
        raise ValueError(f"invalid sidecar {sidecar_path(path)}: fields {sorted(set(payload) ^ set(SIDECAR_FIELDS))}")
    if payload["artifact_contract_version"] != ARTIFACT_CONTRACT_VERSION or payload["rng_scheme_version"] != RNG_SCHEME_VERSION:
        raise ValueError("sidecar artifact contract / RNG scheme is stale or unsupported")
    if payload["method_contract"].get("procedure") != payload["operation"]:
        raise ValueError("method_contract procedure must equal the sidecar operation identifier")
    if payload["artifact_name"] != path.name:
        raise ValueError(f"sidecar artifact_name {payload['artifact_name']!r} does not match {path.name!r}")
    if payload["artifact_size_bytes"] != path.stat().st_size:
        raise ValueError(f"artifact size does not match sidecar: {path}")
    if payload["artifact_sha256"] != sha256_file(path):
        raise ValueError(f"artifact content hash does not match sidecar: {path}")
    for key, wanted in (expected or {}).items():
        if payload.get(key) != wanted:
            raise ValueError(f"incompatible sidecar for {path}: {key}={payload.get(key)!r}, expected {wanted!r}")
    return payload
