"""Artifact-contract version 3 for the simulation stage: authenticated sidecars, immutable shard manifests and the authenticated
``simulation.done.json`` — what the reference's ``analyze ingest`` requires of a simulation tree
(``src/farkle/analysis/ingest.py:182-330``: ``simulation_is_complete`` -> ``resolve_v3_stage_state``, ``load_immutable_manifest_sidecar``,
``validate_authenticated_artifact_metadata`` per row shard, ``compute_manifest_root``).

The reference builds these documents from frozen dataclasses (``src/farkle/utils/authenticated_contract.py``) that a translation layer
(``src/farkle/utils/release_identity.py``) fills from the legacy sidecar metadata of ``simulation/runner.py:338-376``.  Every identity is
the SHA-256 of a canonical JSON text (sorted keys, compact separators, UTF-8; ``canonical_json_bytes`` :99-108), so the documents are
plain nested mappings here, built field for field:

    arrow schema        ArrowSchemaIdentity / _arrow_field_identity           authenticated_contract.py:229-300
    stage config        StageConfigIdentity over the simulation cache scope   :319-361, analysis/stage_registry.py:338-367
    versions            VersionIdentity / release_identity._method_versions   :465-496, release_identity.py:158-281
    method contract     MethodContract / release_identity._typed_method       :499-613, release_identity.py:284-364
    stage identity      StageIdentity / make_stage_identity                   :616-703
    artifact, sources   ArtifactIdentity, SourceArtifactIdentity              :799-843, release_identity._source_role :367-372
    sidecar             AuthenticatedSidecar / make_authenticated_sidecar     :975-1053
    manifest            ManifestEntry, compute_manifest_root, ImmutableManifestSidecar   :846-972, runner._publish_simulation_manifest_v3 :539-636
    completion          AuthenticatedCompletion / release_identity._completion_contract  :1939-1994, release_identity.py:1000-1224

The CODE identity inside every stage identity is the consumer's: ``simulation_is_complete`` recomputes the stage identity with the
identity of the checkout that runs ``analyze ingest`` and compares (``classify_authenticated_lifecycle`` :2044, :2108-2112).  An
independent engine cannot know it, so the caller supplies it — ``commit[:dirty_fingerprint]`` or the path of the reference checkout,
resolved with the same Git commands as ``resolve_code_identity`` (:408-462) — exactly like the reference's own API
(``cfg._code_identity``).  ``oracle/gen_contract_v3.py`` runs the reference's validators and its ingest source snapshot over a
standalone ``farkle run`` tree and pins this module byte for byte against the reference's own writers.
"""
from __future__ import annotations

import hashlib
import json
import os
import re
import subprocess
from pathlib import Path
from typing import Any, Iterable, Mapping, Sequence

ARTIFACT_CONTRACT_VERSION = 3
LIFECYCLE_CONTRACT_VERSION = 1
MANIFEST_CONTRACT_VERSION = 1
OUTCOME_SCHEMA_VERSION = 2
TOURNAMENT_METHOD_VERSION = 2
STAGE_KEY = "simulation"
SCOPE = "diagnostics"
SIMULATION_CACHE_KEY_VERSION = 4  # analysis/stage_registry.py:341
SIDECAR_SUFFIX = ".sidecar.json"
# analysis/stage_registry.py:342-366 — the public configuration that decides what a simulation stage computes
SIMULATION_CACHE_SCOPE = (
    "sim.n_players_list", "sim.seed", "sim.seed_list", "sim.expanded_metrics", "sim.row_dir", "sim.metric_chunk_dir", "sim.score_thresholds",
    "sim.dice_thresholds", "sim.smart_five_opts", "sim.smart_one_opts", "sim.consider_score_opts", "sim.consider_dice_opts",
    "sim.auto_hot_dice_opts", "sim.run_up_score_opts", "sim.include_stop_at", "sim.include_stop_at_heuristic", "screening.resolution_delta",
    "screening.interval_confidence", "screening.max_shuffles_per_root_k", "batching.target_batches", "batching.min_shuffles_per_batch", "rng",
    "artifact_contract")
ARTIFACT_CONTRACT_DEFAULTS = {  # config.py:207-218 (ArtifactContractConfig)
    "artifact_contract_version": 3, "estimand_version": 2, "schema_version": 2, "baseline_version": 1, "k_support_version": 1,
    "weighting_version": 1, "conditioning_version": 2, "multiplicity_version": 1, "candidate_family_version": 1}
_GLOBAL_VERSION_KEYS = {"artifact_contract_version", "rng_scheme_version", "outcome_schema_version", "schema_version", "estimand_version",
                        "conditioning_version"}  # release_identity.py:60-67
OPERATIONS = {  # artifact kind -> operation identifier (simulation/runner.py:424, 1438, 1477, 1488, 1499, 1649, 1708; :629)
    "strategy_manifest": "publish_strategy_manifest", "workload_plan": "publish_simulation_workload_plan",
    "checkpoint": "publish_simulation_checkpoint", "row_shard": "publish_simulation_row_shard",
    "metric_chunk": "publish_simulation_metric_chunk", "shard_manifest": "publish_simulation_shard_manifest",
    "checkpoint_summary": "publish_simulation_checkpoint_summary", "metrics_summary": "publish_simulation_metrics_summary"}
_MEDIA_TYPES = {".json": "application/json", ".jsonl": "application/x-ndjson", ".md": "text/markdown; charset=utf-8",
                ".txt": "text/plain; charset=utf-8", ".png": "image/png", ".npy": "application/x-npy",
                ".pkl": "application/x-python-pickle", ".yaml": "application/yaml", ".yml": "application/yaml"}  # release_identity.py:579-589
_SHA256_RE = re.compile(r"[0-9a-f]{64}")
_PLAIN_RE = re.compile(r"[A-Za-z0-9_./-]+")  # strings whose JSON form is the string in quotes
_PLAIN_LINES_RE = re.compile(r"[A-Za-z0-9_./-]+(?:\n[A-Za-z0-9_./-]+)*")  # ... one per line
_HEX_RUN_RE = re.compile(r"[0-9a-f]*")


class ContractError(ValueError):
    """An artifact tree does not satisfy artifact-contract version 3 (the reference raises ArtifactContractError /
    AuthenticatedContractError, both RuntimeError subclasses; here a ValueError so the CLI reports it like other input errors)."""


# ---- canonical JSON ---------------------------------------------------------------------------------------------------------

def canonical_json_bytes(value: Any) -> bytes:
    """authenticated_contract.py:99-108 (tuples serialise as lists; NaN / infinity are refused)."""
    return json.dumps(value, sort_keys=True, separators=(",", ":"), ensure_ascii=False, allow_nan=False).encode("utf-8")


def _canonical_manifest_text(text: str) -> tuple[dict[str, list[str]], list[str]] | None:
    """A JSON-lines manifest whose every line is ALREADY the canonical JSON of its record (sorted keys, no white space, integers and plain
    strings only, no ``pid`` / ``ts``): the values as text by key, and the lines — or None.  The first line is parsed and re-encoded; the
    others must have its exact shape (one regular expression built from it: the same keys in the same order, a canonical integer where it
    has an integer, a plain string where it has a string), which makes them equal to ``canonical_json_bytes`` of themselves."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    if not lines:
        return None
    try:
        first = json.loads(lines[0])
    except json.JSONDecodeError:
        return None
    if not isinstance(first, dict) or "pid" in first or "ts" in first or canonical_json_bytes(first).decode("utf-8") != lines[0]:
        return None
    parts, keys = [], sorted(first)
    for key in keys:
        value = first[key]
        if not _PLAIN_RE.fullmatch(key) or isinstance(value, bool):
            return None
        if isinstance(value, int):
            parts.append(f'"{re.escape(key)}":(0|-?[1-9][0-9]*)')
        elif isinstance(value, str):
            parts.append(f'"{re.escape(key)}":"([A-Za-z0-9_./-]*)"')
        else:
            return None
    shape = re.compile(r"\{" + ",".join(parts) + r"\}")
    columns: dict[str, list[str]] = {key: [] for key in keys}
    rows = []
    for line in lines:
        m = shape.fullmatch(line)
        if m is None:
            return None
        rows.append(m.groups())
    for key, column in zip(keys, zip(*rows)):
        columns[key] = list(column)
    return columns, lines


def identity_sha256(value: Any) -> str:
    return hashlib.sha256(canonical_json_bytes(value)).hexdigest()


def sha256_bytes(data: bytes) -> str:
    return hashlib.sha256(data).hexdigest()


def sha256_file(path: Path | str) -> str:
    digest = hashlib.sha256()
    with open(path, "rb") as fh:
        while chunk := fh.read(1 << 20):
            digest.update(chunk)
    return digest.hexdigest()


def sidecar_path(path: Path | str) -> Path:
    path = Path(path)
    return path.with_name(path.name + SIDECAR_SUFFIX)


def _atomic_write(path: Path, data: bytes) -> None:
    tmp = path.with_name(f"._tmp_{path.name}")  # (a staging prefix the reference's resume cleanup knows, runner.py:463-469)
    with open(tmp, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


# ---- code identity -------------------------------------------------------------------------------------------------------------

def make_code_identity(commit: str, dirty_fingerprint_sha256: str | None = None, policy: str | None = None) -> dict[str, Any]:
    """CodeIdentity (authenticated_contract.py:371-391): a full Git commit and either a clean tree or the fingerprint of a dirty one."""
    commit = str(commit).strip().lower()
    if not re.fullmatch(r"[0-9a-f]{40}", commit):
        raise ContractError("code identity: commit must be a full lowercase 40-character Git SHA")
    if dirty_fingerprint_sha256 is not None and not _SHA256_RE.fullmatch(dirty_fingerprint_sha256):
        raise ContractError("code identity: the dirty fingerprint must be a lowercase SHA-256 digest")
    state = "clean" if dirty_fingerprint_sha256 is None else "development_dirty"
    policy = policy or ("release_clean" if state == "clean" else "development_dirty")
    if policy not in ("release_clean", "development_dirty"):
        raise ContractError(f"code identity: unknown policy {policy!r}")
    return {"commit": commit, "policy": policy, "state": state, "dirty_fingerprint_sha256": dirty_fingerprint_sha256}


def parse_code_identity(text: str) -> dict[str, Any]:
    """``COMMIT``, ``COMMIT:DIRTY_SHA256`` or ``COMMIT:DIRTY_SHA256:POLICY`` (the value of ``farkle run --code-identity``)."""
    parts = str(text).strip().split(":")
    if not 1 <= len(parts) <= 3:
        raise ContractError(f"code identity {text!r}: expected COMMIT[:DIRTY_SHA256[:POLICY]]")
    return make_code_identity(parts[0], parts[1] if len(parts) > 1 and parts[1] else None, parts[2] if len(parts) > 2 else None)


def resolve_code_identity(repo_root: Path | str, policy: str = "development_dirty",
                          untracked_inventory: Sequence[str] = ("src", "tests", "configs", "pyproject.toml")) -> dict[str, Any]:
    """The identity ``resolve_code_identity`` (authenticated_contract.py:408-462) derives for a checkout: HEAD, and for a dirty tree the
    SHA-256 over the staged diff, the worktree diff and the inventoried untracked files — the value the reference's ``analyze ingest``
    will compute when it runs from that checkout."""
    def git(*args: str) -> bytes:
        try:
            return subprocess.run(["git", *args], cwd=top, check=True, capture_output=True).stdout
        except (OSError, subprocess.CalledProcessError) as exc:
            raise ContractError(f"unable to determine the Git code identity of {repo_root}: {exc}") from exc

    top = Path(repo_root).resolve()
    top = Path(git("rev-parse", "--show-toplevel").decode().strip()).resolve()
    commit = git("rev-parse", "HEAD").decode().strip().lower()
    status = git("status", "--porcelain=v1", "-z", "--untracked-files=all")
    if not status:
        return {"commit": commit, "policy": policy, "state": "clean", "dirty_fingerprint_sha256": None}
    digest = hashlib.sha256()
    digest.update(b"tracked-index\0")
    digest.update(git("diff", "--cached", "--binary", "--no-ext-diff"))
    digest.update(b"tracked-worktree\0")
    digest.update(git("diff", "--binary", "--no-ext-diff"))
    roots = tuple((top / item).resolve() for item in untracked_inventory)
    for raw in sorted(p for p in git("ls-files", "--others", "--exclude-standard", "-z").split(b"\0") if p):
        relative = Path(os.fsdecode(raw))
        absolute = (top / relative).resolve()
        if not any(absolute == root or root in absolute.parents for root in roots):
            continue
        digest.update(b"untracked\0")
        digest.update(relative.as_posix().encode("utf-8"))
        digest.update(b"\0")
        digest.update(sha256_file(absolute).encode("ascii"))
    return {"commit": commit, "policy": policy, "state": "development_dirty", "dirty_fingerprint_sha256": digest.hexdigest()}


# ---- small identities ------------------------------------------------------------------------------------------------------------

def location(relative_path: str) -> dict[str, Any]:
    """CanonicalArtifactLocation of a simulation artifact: relative to the results root, diagnostics scope, no player count (:117-145)."""
    rel = Path(relative_path)
    if not relative_path or rel.is_absolute() or ".." in rel.parts:
        raise ContractError(f"relative_path must remain within the results root: {relative_path!r}")
    return {"stage_key": STAGE_KEY, "scope": SCOPE, "relative_path": rel.as_posix(), "player_count": None}


def _arrow_field(field) -> dict[str, Any]:
    import pyarrow as pa

    dtype = field.type
    children: list = []
    if pa.types.is_struct(dtype) or pa.types.is_union(dtype):
        children = [_arrow_field(dtype.field(i)) for i in range(dtype.num_fields)]
    elif pa.types.is_list(dtype) or pa.types.is_large_list(dtype) or pa.types.is_fixed_size_list(dtype):
        children = [_arrow_field(dtype.value_field)]
    elif pa.types.is_map(dtype):
        children = [_arrow_field(dtype.key_field), _arrow_field(dtype.item_field)]
    return {"name": field.name, "type": str(dtype), "nullable": field.nullable, "children": children}


def arrow_schema_identity(schema, schema_version: int = 2) -> dict[str, Any]:
    fields = [_arrow_field(f) for f in schema]
    return {"schema_version": schema_version, "fields": fields,
            "fingerprint_sha256": identity_sha256({"schema_version": schema_version, "fields": fields})}


def _json_shape(value: Any) -> Any:
    """release_identity.py:624-639 / authenticated_contract.py:725-740: the type skeleton of a JSON document."""
    if isinstance(value, dict):
        return {str(k): _json_shape(v) for k, v in sorted(value.items())}
    if isinstance(value, list):
        return [_json_shape(value[0])] if value else []
    if value is None:
        return "null"
    if isinstance(value, bool):
        return "boolean"
    if isinstance(value, int):
        return "integer"
    if isinstance(value, float):
        return "number"
    if isinstance(value, str):
        return "string"
    raise ContractError(f"unsupported JSON value {type(value).__name__}")


def format_identity(name: str, data: bytes) -> dict[str, Any] | None:
    """ArtifactFormatIdentity of a non-Parquet artifact (release_identity._format_identity :575-621); None for Parquet."""
    suffix = Path(name).suffix.lower()
    if suffix == ".parquet":
        return None
    structural = None
    if suffix == ".json":
        structural = identity_sha256(_json_shape(json.loads(data.decode("utf-8"))))
    elif suffix == ".jsonl":
        text = data.decode("utf-8")
        try:
            shapes = {json.dumps(_json_shape(json.loads(text)), sort_keys=True)}
        except json.JSONDecodeError:
            shapes = {json.dumps(_json_shape(json.loads(line)), sort_keys=True) for line in text.splitlines() if line.strip()}
        structural = identity_sha256(sorted(shapes))
    return {"media_type": _MEDIA_TYPES.get(suffix, "application/octet-stream"), "format_version": 1, "structural_schema_sha256": structural}


def effective_simulation_config(cfg) -> dict[str, Any]:
    """The part of ``effective_config_dict`` (config.py:1740-1751) the simulation cache scope selects: dataclass sections as mappings,
    paths as strings.  ``artifact_contract`` is a section this package carries opaquely: the reference's defaults under what it sets."""
    import dataclasses

    def plain(obj: Any) -> Any:
        if isinstance(obj, Path):
            return str(obj)
        if isinstance(obj, Mapping):
            return {k: plain(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return [plain(v) for v in obj]
        return obj

    contract = dict(ARTIFACT_CONTRACT_DEFAULTS)
    for key, value in (cfg.opaque.get("artifact_contract") or {}).items():
        if key not in contract:
            raise ContractError(f"Unknown option {key!r} in config section 'artifact_contract'")
        contract[key] = int(value)
    sim = dataclasses.asdict(cfg.sim)
    return plain({"sim": sim, "screening": dataclasses.asdict(cfg.screening), "batching": dataclasses.asdict(cfg.batching),
                  "rng": {"scheme_version": cfg.rng.scheme_version, "bit_generator": cfg.rng.bit_generator}, "artifact_contract": contract})


def _flatten(value: Mapping[str, Any], prefix: str) -> list[str]:
    out: list[str] = []
    for key, item in sorted(value.items()):
        path = f"{prefix}.{key}"
        out.extend(_flatten(item, path) if isinstance(item, Mapping) else [path])
    return out


def stage_config_identity(cfg) -> dict[str, Any]:
    """stage_config_identity over release_identity._stage_field_paths (:106-127): scoped sections expand to their leaves."""
    public = effective_simulation_config(cfg)
    paths: list[str] = []
    for scoped in SIMULATION_CACHE_SCOPE:
        cursor: Any = public
        for part in scoped.split("."):
            cursor = cursor[part]
        paths.extend(_flatten(cursor, scoped) if isinstance(cursor, Mapping) else [scoped])
    paths = sorted(set(paths))
    selected = {}
    for path in paths:
        cursor = public
        for part in path.split("."):
            cursor = cursor[part]
        selected[path] = cursor
    return {"stage_key": STAGE_KEY, "field_paths": paths, "selected_config": selected,
            "sha256": identity_sha256({"stage_key": STAGE_KEY, "field_paths": paths, "selected_config": selected})}


def version_identity(cfg, parameters: Mapping[str, Any] | None = None) -> tuple[dict[str, Any], int]:
    """release_identity._versions / _method_versions (:158-281) for the simulation stage -> (VersionIdentity, method_version)."""
    contract = effective_simulation_config(cfg)["artifact_contract"]
    accepted = (contract["artifact_contract_version"], cfg.rng.scheme_version, OUTCOME_SCHEMA_VERSION, contract["schema_version"],
                contract["estimand_version"], contract["conditioning_version"])
    if accepted != (3, 2, 2, 2, 2, 2):
        raise ContractError(f"incomplete authenticated-v3 version identity: {accepted}; expected (3, 2, 2, 2, 2, 2)")
    versions: dict[str, int] = {"tournament_method_version": TOURNAMENT_METHOD_VERSION}  # AppConfig.freshness_key (config.py:527-573)
    for key in ("baseline_version", "k_support_version", "weighting_version", "multiplicity_version", "candidate_family_version"):
        versions[key] = int(contract[key])
    for key, value in (parameters or {}).items():
        if key.endswith("_version") and key not in _GLOBAL_VERSION_KEYS and isinstance(value, int) and not isinstance(value, bool):
            versions[key] = int(value)
    explicit = (parameters or {}).get("method_version")
    if isinstance(explicit, int) and not isinstance(explicit, bool):
        method_version = explicit
    else:
        candidates = [v for k, v in versions.items() if k.startswith(STAGE_KEY) and k.endswith("_method_version")]
        method_version = candidates[0] if candidates else 1
    versions.setdefault(f"{STAGE_KEY}_operation_method_version", method_version)
    return ({"artifact_contract_version": 3, "lifecycle_contract_version": LIFECYCLE_CONTRACT_VERSION, "rng_scheme_version": 2,
             "outcome_schema_version": 2, "schema_version": 2, "estimand_version": 2, "conditioning_version": 2,
             "method_versions": dict(sorted(versions.items()))}, method_version)


def _root_seeds(cfg) -> list[int]:
    return sorted({int(v) for v in (cfg.sim.seed_list or [cfg.sim.seed])})


def method_contract(cfg, *, procedure: str, method_version: int, baseline: str, replication_unit: str, conditioning: str = "unconditional",
                    source_scope: str | None = None, player_counts: Sequence[int] = (), weighted_quantity: str = "none",
                    support_count_role: str = "raw_support_provenance", uncertainty_method: str = "none", missing_cell_policy: str = "not_applicable",
                    multiplicity: str | None = None, semantic_contract: Mapping[str, Any] | None = None) -> dict[str, Any]:
    """MethodContract (authenticated_contract.py:499-613) with the fields a simulation artifact sets (release_identity._typed_method)."""
    counts = sorted({int(v) for v in player_counts})
    return {
        "procedure": procedure, "method_version": method_version, "baseline": baseline, "replication_unit": replication_unit, "k_weights": None,
        "multiplicity": multiplicity, "family_hash": None, "schedule_hash": None, "practical_margin": None, "equivalence_margin": None,
        "ordinary_alpha": None, "simultaneous_alpha": None, "conditioning": conditioning, "source_scope": source_scope,
        "root_seeds": _root_seeds(cfg), "player_counts": counts, "required_player_counts": counts, "weighted_quantity": weighted_quantity,
        "support_count_role": support_count_role, "uncertainty_method": uncertainty_method, "k_aggregation_method": "none",
        "missing_cell_policy": missing_cell_policy, "seed_scope": "single_root", "consistency_columns": [], "grouping_keys": [],
        "semantic_contract_sha256": None if semantic_contract is None else identity_sha256(semantic_contract),
        "rng_effective_matchup_group_cap": None, "rng_diagnostic_lags": [], "rng_tracked_matchup_group_count": None,
        "rng_skipped_matchup_group_count": None, "rng_skipped_matchup_row_count": None}


def make_stage_identity(*, stage_config: Mapping[str, Any], versions: Mapping[str, Any], code: Mapping[str, Any], method: Mapping[str, Any],
                        upstream: Sequence[str], designs: Mapping[str, str]) -> dict[str, Any]:
    """make_stage_identity (authenticated_contract.py:659-702)."""
    if method["method_version"] not in versions["method_versions"].values():
        raise ContractError("method contract version is absent from version identity")
    method_sha = identity_sha256(method)
    designs = dict(sorted(designs.items()))
    provisional = {"lifecycle_contract_version": versions["lifecycle_contract_version"], "stage_key": STAGE_KEY,
                   "stage_cache_key_version": SIMULATION_CACHE_KEY_VERSION, "stage_config_identity": stage_config, "versions": versions,
                   "code_identity": code, "method_contract_sha256": method_sha, "upstream_identities": list(upstream),
                   "immutable_design_identities": designs}
    return {"stage_key": STAGE_KEY, "stage_cache_key_version": SIMULATION_CACHE_KEY_VERSION, "stage_config": stage_config, "versions": versions,
            "code": code, "method_contract_sha256": method_sha, "upstream_identity_sha256": list(upstream),
            "immutable_design_identities": designs, "sha256": identity_sha256(provisional)}


def source_role(sidecar: Mapping[str, Any]) -> str:
    """release_identity._source_role (:367-372)."""
    loc = sidecar["artifact"]["location"]
    roots = "_".join(str(v) for v in sidecar["method_contract"]["root_seeds"]) or "none"
    k = "all" if loc["player_count"] is None else str(loc["player_count"])
    relative = loc["relative_path"].replace("/", ".").replace("\\", ".")
    return f"artifact.{loc['stage_key']}.{loc['scope']}.k_{k}.roots_{roots}.{relative}"


def manifest_role(loc: Mapping[str, Any]) -> str:
    """release_identity._manifest_role (:375-379)."""
    k = "all" if loc["player_count"] is None else str(loc["player_count"])
    relative = loc["relative_path"].replace("/", ".").replace("\\", ".")
    return f"manifest.{loc['stage_key']}.{loc['scope']}.k_{k}.{relative}"


def manifest_entry(shuffle_or_block: int, canonical_relative_path: str, data_sha256: str, sidecar_sha256: str, schema_fingerprint_sha256: str) -> dict:
    return {"coordinate": [int(shuffle_or_block)], "canonical_relative_path": canonical_relative_path, "data_sha256": data_sha256,
            "sidecar_sha256": sidecar_sha256, "schema_fingerprint_sha256": schema_fingerprint_sha256}


def compute_manifest_root(entries: Iterable[Mapping[str, Any]]) -> dict[str, Any]:
    """compute_manifest_root (authenticated_contract.py:882-905): length-prefixed canonical entries, strictly increasing coordinates."""
    root, support = hashlib.sha256(), hashlib.sha256()
    previous = None
    count = 0
    for entry in entries:
        coordinate = tuple(entry["coordinate"])
        if previous is not None and coordinate <= previous:
            raise ContractError("manifest entries must have strictly increasing coordinates")
        encoded = canonical_json_bytes(entry)
        root.update(len(encoded).to_bytes(8, "big"))
        root.update(encoded)
        key = canonical_json_bytes(list(coordinate))
        support.update(len(key).to_bytes(8, "big"))
        support.update(key)
        previous = coordinate
        count += 1
    return {"root_sha256": root.hexdigest(), "coordinate_support_sha256": support.hexdigest(), "entry_count": count}


# ---- the simulation stage's writer -------------------------------------------------------------------------------------------------

class SimulationContract:
    """Artifact-contract-v3 documents of one results root (``cfg``), under one code identity.

    ``sidecar_bytes`` / ``write_sidecar`` bind an artifact that is complete on disk (or whose bytes the caller holds);
    ``shard_template`` hands the row-shard writer processes the constant text of a shard's sidecar, so that a shard costs two SHA-256
    passes and one string join; ``publish_manifest`` seals a shard manifest; ``write_completion`` publishes the stamp last."""

    def __init__(self, cfg, code_identity: Mapping[str, Any], *, game_profile_sha256: str | None = None, run_lineage_sha256: str | None = None):
        self.cfg = cfg
        self.root = Path(cfg.results_root)
        self.code = dict(code_identity)
        self.stage_config = stage_config_identity(cfg)
        self.designs: dict[str, str] = {}
        if run_lineage_sha256 is not None:
            self.designs["run_lineage_sha256"] = run_lineage_sha256
        if game_profile_sha256 is not None:
            self.designs["game_profile_sha256"] = game_profile_sha256
        self._sources: dict[str, dict] = {}   # path -> SourceArtifactIdentity (a published input is captured once)

    # -- identities of inputs ----
    def relative(self, path: Path | str) -> str:
        try:
            return Path(os.path.abspath(path)).relative_to(os.path.abspath(self.root)).as_posix()
        except ValueError as exc:
            raise ContractError(f"artifact {path} is outside the canonical simulation root {self.root}") from exc

    def capture_source(self, path: Path | str) -> dict[str, Any]:
        """SourceArtifactIdentity of an authenticated artifact of this tree (capture_source_artifact_unbound :1567-1580; the bytes are
        checked against the sidecar once per run)."""
        key = os.fspath(path)
        if key not in self._sources:
            side = sidecar_path(path)
            try:
                raw = side.read_bytes()
            except FileNotFoundError as exc:
                raise ContractError(f"v3 source artifact is missing an authenticated sidecar: {path}") from exc
            doc = json.loads(raw)
            if doc.get("artifact_contract_version") != 3 or "manifest_contract_version" in doc:
                raise ContractError(f"contract-v2 source cannot satisfy a v3 publication: {path}")
            art = doc["artifact"]
            if os.stat(path).st_size != art["byte_length"] or sha256_file(path) != art["content_sha256"]:
                raise ContractError(f"source artifact bytes do not match its sidecar: {path}")
            self._sources[key] = {"logical_role": source_role(doc), "artifact": art, "sidecar_sha256": sha256_bytes(raw),
                                  "sidecar_contract_sha256": doc["sidecar_contract_sha256"]}
        return self._sources[key]

    # -- one ordinary artifact ----
    def _frame(self, operation: str, player_counts: Sequence[int], sources: Sequence[Path | str]) -> tuple[dict, dict, dict, list]:
        """(method contract, versions, stage identity, source identities) of an output of ``_simulation_output_sidecar`` (runner.py:338-376)."""
        parameters = {"tournament_method_version": TOURNAMENT_METHOD_VERSION, "outcome_schema_version": OUTCOME_SCHEMA_VERSION}
        versions, method_version = version_identity(self.cfg, parameters)
        method = method_contract(
            self.cfg, procedure=operation, method_version=method_version, baseline="tournament_design", replication_unit="shuffle",
            conditioning="all_attempted_games", source_scope=SCOPE, player_counts=player_counts, weighted_quantity="raw_simulation_evidence",
            support_count_role="root_k_simulation", uncertainty_method="deterministic_monte_carlo", missing_cell_policy="fail",
            multiplicity="holm_h2h", semantic_contract={"kind": "operation", "procedure": operation, "parameters": parameters})
        captured = sorted((self.capture_source(p) for p in sources), key=lambda s: s["logical_role"])
        stage = make_stage_identity(stage_config=self.stage_config, versions=versions, code=self.code, method=method,
                                    upstream=[identity_sha256(s) for s in captured], designs=self.designs)
        return method, versions, stage, captured

    @staticmethod
    def _document(artifact: Mapping[str, Any], method: Mapping, versions: Mapping, stage: Mapping, sources: Sequence[Mapping]) -> bytes:
        payload = {"artifact_contract_version": ARTIFACT_CONTRACT_VERSION, "artifact": artifact, "stage_identity": stage, "method_contract": method,
                   "versions": versions, "source_artifacts": list(sources), "manifest_roots": []}
        return canonical_json_bytes({**payload, "sidecar_contract_sha256": identity_sha256(payload)}) + b"\n"

    def sidecar_for_identity(self, relative_path: str, kind: str, *, byte_length: int, content_sha256: str, arrow_schema: Mapping | None = None,
                             format_identity: Mapping | None = None, n_players: int, sources: Sequence[Path | str] = (),
                             support_counts: Sequence[int] | None = None) -> bytes:
        """The sidecar text of an artifact known by its identity: exactly one of ``arrow_schema`` (ArrowSchemaIdentity of a Parquet file)
        and ``format_identity`` (ArtifactFormatIdentity of anything else), as ArtifactIdentity demands (authenticated_contract.py:815-818)."""
        if (arrow_schema is None) == (format_identity is None):
            raise ContractError("artifact identity requires exactly one Arrow schema or non-Parquet format identity")
        if byte_length <= 0:
            raise ContractError(f"artifact writer did not create {relative_path}")
        operation = OPERATIONS[kind]
        artifact = {"location": location(relative_path), "byte_length": int(byte_length), "content_sha256": content_sha256,
                    "arrow_schema": arrow_schema, "logical_operation": operation, "format_identity": format_identity}
        method, versions, stage, captured = self._frame(operation, support_counts if support_counts is not None else [n_players], sources)
        return self._document(artifact, method, versions, stage, captured)

    def sidecar_bytes(self, path: Path | str, kind: str, *, n_players: int, sources: Sequence[Path | str] = (),
                      support_counts: Sequence[int] | None = None, data: bytes | None = None, schema=None) -> bytes:
        """The sidecar text of the artifact at ``path`` (bytes read from disk unless ``data`` holds them; Parquet: ``schema`` = its Arrow
        schema when the caller has it, else read from the file's footer)."""
        name = Path(path).name
        if data is None:
            data = Path(path).read_bytes()
        arrow = fmt = None
        if name.endswith(".parquet"):
            if schema is None:
                import pyarrow as pa
                import pyarrow.parquet as pq

                schema = pq.read_schema(pa.BufferReader(data))
            arrow = arrow_schema_identity(schema)
        else:
            fmt = format_identity(name, data)
        return self.sidecar_for_identity(self.relative(path), kind, byte_length=len(data), content_sha256=sha256_bytes(data), arrow_schema=arrow,
                                         format_identity=fmt, n_players=n_players, sources=sources, support_counts=support_counts)

    def write_sidecar(self, path: Path | str, kind: str, **kw) -> bytes:
        text = self.sidecar_bytes(path, kind, **kw)
        _atomic_write(sidecar_path(path), text)
        return text

    # -- shards: constant text around (byte_length, content_sha256, relative_path) ----
    def shard_template(self, kind: str, directory: Path | str, schema, *, n_players: int, sources: Sequence[Path | str]) -> dict[str, Any]:
        """A picklable template the shard writers fill per file (``fill_shard_template``): the sidecar of ``directory/<name>`` for a Parquet
        shard of ``schema``.  The marker values cannot occur in the constant part (hex digests and an integer)."""
        operation = OPERATIONS[kind]
        method, versions, stage, captured = self._frame(operation, [n_players], sources)
        arrow = arrow_schema_identity(schema)
        marks = {"len": 987_654_321_012_345_678, "sha": "@content_sha256@", "rel": "@relative_path@", "side": "@sidecar_contract_sha256@"}
        artifact = {"location": {**location("x"), "relative_path": marks["rel"]}, "byte_length": marks["len"], "content_sha256": marks["sha"],
                    "arrow_schema": arrow, "logical_operation": operation, "format_identity": None}
        payload = {"artifact_contract_version": ARTIFACT_CONTRACT_VERSION, "artifact": artifact, "stage_identity": stage, "method_contract": method,
                   "versions": versions, "source_artifacts": captured, "manifest_roots": []}

        def split(text: str, names: Sequence[str]) -> list[str]:
            pieces = []
            for name in names:  # canonical key order puts the markers in this order: byte_length, content_sha256, relative_path, digest
                token = json.dumps(marks[name])
                if text.count(token) != 1:
                    raise ContractError("shard sidecar template: a marker value occurs in the constant text")
                head, text = text.split(token)
                pieces.append(head)
            return pieces + [text]

        body = split(canonical_json_bytes(payload).decode("utf-8"), ("len", "sha", "rel"))
        full = split(canonical_json_bytes({**payload, "sidecar_contract_sha256": marks["side"]}).decode("utf-8"), ("len", "sha", "rel", "side"))
        return {"body": body, "full": full, "directory": self.relative(Path(directory) / "x")[:-1],
                "schema_fingerprint_sha256": arrow["fingerprint_sha256"]}

    # -- manifests ----
    def publish_manifest(self, path: Path | str, records: Sequence[Mapping[str, Any]] | None = None, *, n_players: int) -> dict[str, Any]:
        """Seal a shard manifest (runner._publish_simulation_manifest_v3 :539-636 -> publish_native_manifest_v3, release_identity.py:717-814):
        the native file becomes one canonical JSON line per record in coordinate order (process ids and timestamps dropped), and the adjacent
        sidecar binds its SHA-256 and the coordinate-sorted root over the shards' byte / sidecar / schema identities the records carry.
        ``records``: the file's records, parsed; None = read them from ``path`` — where a manifest whose lines are already canonical (what
        this package's writers append) is sealed from its text: 12 ms per 4 300 shards instead of 60."""
        path = Path(path)
        prefix = self.relative(path.parent / "x")[:-1]  # shards sit beside their manifest: one path computation per manifest, not per shard
        coordinates = identities = native = None
        if records is None:
            text = path.read_text(encoding="utf-8")
            sealed = _canonical_manifest_text(text)
            if sealed is not None:
                columns, lines = sealed
                key = next((name for name in ("shuffle_index", "process_block_index", "deterministic_batch_id") if name in columns), None)
                coordinates = [int(v) for v in columns[key]] if key is not None else list(range(len(lines)))
                try:
                    identities = list(zip([prefix + name if "/" not in name and name not in ("", ".", "..") else self.relative(path.parent / name)
                                           for name in columns["path"]], columns["data_sha256"], columns["schema_fingerprint_sha256"],
                                          columns["sidecar_sha256"]))
                except KeyError as exc:
                    raise ContractError(f"simulation manifest record without its shard identity ({exc.args[0]}): {path}") from exc
                order = sorted(range(len(lines)), key=coordinates.__getitem__)
                if order != list(range(len(lines))):
                    coordinates, identities, lines = [coordinates[i] for i in order], [identities[i] for i in order], [lines[i] for i in order]
                native = ("\n".join(lines) + "\n").encode("utf-8")
            else:
                records = []
                for line in text.splitlines():
                    try:
                        records.append(json.loads(line)) if line.strip() else None
                    except json.JSONDecodeError:
                        continue  # a torn last line of an interrupted append
        if coordinates is None:
            keyed = []
            for index, record in enumerate(records):
                coordinate = int(record.get("shuffle_index", record.get("process_block_index", record.get("deterministic_batch_id", index))))
                name = str(record["path"])
                relative = prefix + name if "/" not in name and "\\" not in name and name not in ("", ".", "..") else self.relative(path.parent / name)
                try:
                    identity = (relative, record["data_sha256"], record["schema_fingerprint_sha256"], record["sidecar_sha256"])
                except KeyError as exc:
                    raise ContractError(f"simulation manifest record without its shard identity ({exc.args[0]}): {path}") from exc
                keyed.append((coordinate, identity, record))
            keyed.sort(key=lambda item: item[0])
            coordinates, identities = [item[0] for item in keyed], [item[1] for item in keyed]
            native = b"".join(canonical_json_bytes(record if "pid" not in record and "ts" not in record else
                                                   {k: v for k, v in record.items() if k not in ("pid", "ts")}) + b"\n" for _, _, record in keyed)
        if not coordinates:
            raise ContractError(f"simulation manifest has no authenticated entries: {path}")
        if any(b <= a for a, b in zip(coordinates, coordinates[1:])):
            raise ContractError("manifest entries must have strictly increasing coordinates")
        # compute_manifest_root over ManifestEntry documents; their canonical JSON is written out directly when every string is plain
        # (hex digests and file names: no escaping) — checked in bulk, encoded column-wise and hashed in one update
        digests = [d for identity in identities for d in identity[1:]]
        plain = (all(isinstance(d, str) and len(d) == 64 for d in digests) and _HEX_RUN_RE.fullmatch("".join(digests)) is not None
                 and all(isinstance(identity[0], str) for identity in identities)
                 and _PLAIN_LINES_RE.fullmatch("\n".join(identity[0] for identity in identities)) is not None)
        if plain:
            entries = [(f'{{"canonical_relative_path":"{relative}","coordinate":[{coordinate}],"data_sha256":"{data_sha}",'
                        f'"schema_fingerprint_sha256":"{schema_sha}","sidecar_sha256":"{side_sha}"}}').encode("ascii")
                       for coordinate, (relative, data_sha, schema_sha, side_sha) in zip(coordinates, identities)]
        else:
            entries = [canonical_json_bytes(manifest_entry(coordinate, relative, data_sha, side_sha, schema_sha))
                       for coordinate, (relative, data_sha, schema_sha, side_sha) in zip(coordinates, identities)]
        keys = [b"[%d]" % coordinate for coordinate in coordinates]
        root = hashlib.sha256(b"".join(len(e).to_bytes(8, "big") + e for e in entries))
        support = hashlib.sha256(b"".join(len(key).to_bytes(8, "big") + key for key in keys))
        summary = {"root_sha256": root.hexdigest(), "coordinate_support_sha256": support.hexdigest(), "entry_count": len(coordinates)}
        operation = OPERATIONS["shard_manifest"]
        versions, method_version = version_identity(self.cfg, None)
        method = method_contract(
            self.cfg, procedure=operation, method_version=method_version, baseline="coordinate_identity", replication_unit="manifest_coordinate",
            conditioning="unconditional", source_scope=SCOPE, player_counts=[n_players], weighted_quantity="immutable_shard_inventory",
            support_count_role="coordinate_sorted_shards", uncertainty_method="none", missing_cell_policy="fail", multiplicity="holm_h2h",
            semantic_contract={"kind": "operation", "procedure": operation})
        stage = make_stage_identity(stage_config=self.stage_config, versions=versions, code=self.code, method=method, upstream=[], designs=self.designs)
        payload = {"artifact_contract_version": ARTIFACT_CONTRACT_VERSION, "manifest_contract_version": MANIFEST_CONTRACT_VERSION,
                   "location": location(self.relative(path)), "manifest_sha256": sha256_bytes(native), "summary": summary, "stage_identity": stage}
        document = {**payload, "sidecar_contract_sha256": identity_sha256(payload)}
        sidecar_path(path).unlink(missing_ok=True)  # a crash leaves a manifest without sidecar, never new bytes under an old one
        _atomic_write(path, native)
        _atomic_write(sidecar_path(path), canonical_json_bytes(document) + b"\n")
        return document

    # -- completion ----
    def completion(self, outputs: Sequence[Path | str], state: str = "complete_valid") -> dict[str, Any]:
        """AuthenticatedCompletion over the stage's output inventory (release_identity._completion_contract :1000-1145): every output's exact
        sidecar bytes enter the stage identity as ``output:<relative path>`` design identities; shard directories are represented by their
        sealed manifests."""
        files = sorted({os.path.realpath(p): Path(p) for p in outputs}.items())
        identities, designs, artifact_versions, manifest_versions = [], dict(self.designs), None, None
        for _, path in files:
            side = sidecar_path(path)
            try:
                raw = side.read_bytes()
            except FileNotFoundError as exc:
                raise ContractError(f"authenticated completion requires a valid sidecar for {path}") from exc
            doc = json.loads(raw)
            if not isinstance(doc, dict) or doc.get("artifact_contract_version") != 3:
                raise ContractError(f"contract-v2 artifact cannot satisfy v3 completion: {path}")
            side_sha = sha256_bytes(raw)
            loc = location(self.relative(path))
            if "manifest_contract_version" in doc:
                if doc["location"] != loc:
                    raise ContractError(f"manifest scope/path identity mismatch: {path}")
                root = {"logical_role": manifest_role(loc), "location": loc, "manifest_sha256": doc["manifest_sha256"], "sidecar_sha256": side_sha,
                        "sidecar_contract_sha256": doc["sidecar_contract_sha256"], "summary": doc["summary"]}
                identities.append({"artifact": None, "manifest": root, "sidecar_sha256": side_sha})
                manifest_versions = manifest_versions or doc["stage_identity"]["versions"]
            else:
                if doc["artifact"]["location"] != loc:
                    raise ContractError(f"artifact scope/path identity mismatch: {path}")
                if os.stat(path).st_size != doc["artifact"]["byte_length"]:
                    raise ContractError(f"artifact byte length does not match sidecar: {path}")
                identities.append({"artifact": doc["artifact"], "manifest": None, "sidecar_sha256": side_sha})
                artifact_versions = artifact_versions or doc["versions"]
            designs[f"output:{loc['relative_path']}"] = side_sha
        if not identities:
            raise ContractError("authenticated completion requires stage outputs")
        versions = artifact_versions or manifest_versions  # (sidecars[0].versions if sidecars else the first manifest's, release_identity.py:1105-1109)
        method = method_contract(
            self.cfg, procedure=f"{STAGE_KEY}_authenticated_completion", method_version=next(iter(versions["method_versions"].values())),
            baseline="stage_output_inventory", replication_unit="stage_run", conditioning="authenticated_complete_outputs",
            player_counts=sorted({int(v) for v in self.cfg.sim.n_players_list}))
        stage = make_stage_identity(stage_config=self.stage_config, versions=versions, code=self.code, method=method, upstream=[], designs=designs)
        identities.sort(key=lambda item: canonical_json_bytes((item["artifact"] or item["manifest"])["location"]))
        return {"lifecycle_contract_version": LIFECYCLE_CONTRACT_VERSION, "stage_identity_sha256": stage["sha256"], "state": state, "outputs": identities}

    def write_completion(self, done_path: Path | str, outputs: Sequence[Path | str]) -> dict[str, Any]:
        completion = self.completion(outputs)
        _atomic_write(Path(done_path), canonical_json_bytes(completion) + b"\n")
        return completion

    def is_complete(self, done_path: Path | str) -> bool:
        """``simulation_is_complete`` for a v3 stamp (runner.py:274-317 -> resolve_v3_stage_state, the metadata-level check the reference
        uses for the simulation stage): the stamp's outputs exist with the recorded sidecars and byte lengths, and the stage identity
        recomputed from them under the CURRENT configuration and code identity equals the stamp's."""
        try:
            stamp = json.loads(Path(done_path).read_text(encoding="utf-8"))
            outputs = [self.root / (item.get("artifact") or item.get("manifest"))["location"]["relative_path"] for item in stamp["outputs"]]
            current = self.completion(outputs, state=stamp.get("state", "complete_valid"))
        except (OSError, KeyError, TypeError, ValueError):
            return False
        if stamp.get("state") != "complete_valid" or current["stage_identity_sha256"] != stamp.get("stage_identity_sha256"):
            return False
        if current["outputs"] != stamp["outputs"]:
            return False
        for item in stamp["outputs"]:  # sealed manifests still hash to what their sidecar binds
            if item["manifest"] is not None:
                if sha256_file(self.root / item["manifest"]["location"]["relative_path"]) != item["manifest"]["manifest_sha256"]:
                    return False
        return True


def fill_shard_template(template: Mapping[str, Any], name: str, byte_length: int, content_sha256: str) -> tuple[bytes, str]:
    """(sidecar bytes, SHA-256 of the sidecar file) of one shard from ``SimulationContract.shard_template``: the contract digest is the
    SHA-256 of the canonical payload without the digest member, the file is the canonical payload with it plus a newline."""
    b, f = template["body"], template["full"]
    length, sha, rel = str(int(byte_length)), f'"{content_sha256}"', json.dumps(template["directory"] + name, ensure_ascii=False)
    digest = hashlib.sha256("".join((b[0], length, b[1], sha, b[2], rel, b[3])).encode("utf-8")).hexdigest()
    text = "".join((f[0], length, f[1], sha, f[2], rel, f[3], f'"{digest}"', f[4], "\n")).encode("utf-8")
    return text, hashlib.sha256(text).hexdigest()
