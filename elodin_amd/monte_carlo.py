"""Campaign plan for GPU Monte-Carlo: rollouts become rows of the entity axis.

Restates the plan semantics of the reference's sampler
(libs/nox-py/python/elodin/monte_carlo/sample.py:84-151) so that a GPU campaign runs the SAME
plan a process-per-rollout campaign would: Latin-hypercube (or plain random) unit samples drawn
from `random.Random(seed)` column by column, mapped through fixed / choice / uniform / loguniform /
normal, crossed with `sim_sweep` and `meta_sweep` grids (sim x meta x mc order), rows numbered
`run_id = run_%07d`, `seed = idx + 1` (sample.py:149; PlanRow libs/monte-carlo/src/lib.rs:263-268).
Instead of a CSV of strings it yields the dense float64 parameter table the kernels consume
(one row per rollout, columns in sorted key order) plus the same CSV text for interchange.
"""
from __future__ import annotations

import io
import itertools
import math
import random
from dataclasses import dataclass, field
from statistics import NormalDist
from typing import Any, Dict, List, Mapping, Optional, Sequence

import numpy as np

DISTRIBUTIONS = ("fixed", "choice", "uniform", "loguniform", "normal")
_LO_KEYS, _HI_KEYS = ("min", "lo", "low"), ("max", "hi", "high")


def _first(spec: Mapping[str, Any], keys: Sequence[str]):
    return next((spec[k] for k in keys if k in spec), None)


@dataclass(frozen=True)
class Variable:
    name: str
    dist: str
    spec: Mapping[str, Any]

    @staticmethod
    def parse(name: str, spec: Any) -> "Variable":
        if not isinstance(spec, Mapping):
            raise ValueError(f'variable "{name}" must be a table like {{ dist = "normal", ... }}')
        dist = str(spec.get("dist", "fixed")).lower()
        given = ", ".join(sorted(k for k in spec if k != "dist")) or "nothing"
        if dist not in DISTRIBUTIONS:
            raise ValueError(f'unknown dist "{dist}" for "{name}" (known: {", ".join(DISTRIBUTIONS)})')
        lo, hi = _first(spec, _LO_KEYS), _first(spec, _HI_KEYS)
        need = None
        if dist == "fixed" and "value" not in spec:
            need = "value"
        elif dist == "choice" and not spec.get("values"):
            need = "a non-empty values list"
        elif dist in ("uniform", "loguniform") and (lo is None or hi is None):
            need = "min/max"
        elif dist == "normal" and not ("mean" in spec and "std" in spec):
            need = "mean/std"
        if need:
            raise ValueError(f'{dist} for "{name}" needs {need} (got: {given})')
        if dist == "loguniform" and (float(lo) <= 0 or float(hi) <= 0):
            raise ValueError(f'loguniform for "{name}" needs positive min/max')
        return Variable(name, dist, spec)

    def at(self, u: float):
        """Inverse-CDF style map of a unit sample u in [0,1)."""
        s = self.spec
        if self.dist == "fixed":
            return s.get("value")
        if self.dist == "choice":
            vals = s["values"]
            return vals[min(int(u * len(vals)), len(vals) - 1)]
        if self.dist == "normal":
            u = min(max(u, 1e-12), 1.0 - 1e-12)
            return float(s["mean"]) + float(s["std"]) * NormalDist().inv_cdf(u)
        lo, hi = float(_first(s, _LO_KEYS)), float(_first(s, _HI_KEYS))
        if self.dist == "uniform":
            return lo + (hi - lo) * u
        llo, lhi = math.log(lo), math.log(hi)
        return math.exp(llo + (lhi - llo) * u)


def _unit_samples(n: int, d: int, method: str, rng: random.Random) -> List[List[float]]:
    if method == "random":
        return [[rng.random() for _ in range(d)] for _ in range(n)]
    # Latin hypercube: per column, one jittered sample per stratum, then a shuffle of that column
    table = [[0.0] * d for _ in range(n)]
    for c in range(d):
        strata = [(k + rng.random()) / n for k in range(n)]
        rng.shuffle(strata)
        for r in range(n):
            table[r][c] = strata[r]
    return table


def _grid(table: Mapping[str, Sequence[Any]] | None, prefix: str) -> List[Dict[str, Any]]:
    if not table:
        return [{}]
    names = sorted(table)
    return [{f"{prefix}.{k}": v for k, v in zip(names, combo)}
            for combo in itertools.product(*(table[k] for k in names))]


@dataclass
class Plan:
    headers: List[str]                 # "run_id", "seed", then sorted param./meta. keys
    rows: List[Dict[str, Any]]         # one dict per rollout, in run order
    run_ids: List[str] = field(default_factory=list)
    seeds: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.uint64))

    def __len__(self) -> int:
        return len(self.rows)

    @property
    def param_names(self) -> List[str]:
        return [h[len("param."):] for h in self.headers if h.startswith("param.")]

    def table(self, names: Sequence[str] | None = None, defaults: Mapping[str, float] | None = None) -> np.ndarray:
        """Dense float64 [n_runs, n_params] matrix (the thing broadcast to every GPU)."""
        names = list(names) if names is not None else self.param_names
        defaults = defaults or {}
        out = np.empty((len(self.rows), len(names)), dtype=np.float64)
        for r, row in enumerate(self.rows):
            for c, name in enumerate(names):
                v = row.get(f"param.{name}", defaults.get(name))
                if v is None:
                    raise KeyError(f'parameter "{name}" is neither in the plan nor in defaults')
                out[r, c] = float(v)
        return out

    def context(self, index: int, run_dir: Optional[str] = None, **fields) -> Dict[str, Any]:
        """The per-run context document the reference's runner writes for rollout `index` and points
        $ELODIN_MONTE_CARLO_CONTEXT at (libs/nox-py/src/monte_carlo.rs:29-43 `ContextData`; the row -> params / meta split
        and the cell parsing are `read_plan`'s, libs/monte-carlo/src/lib.rs:2615-2661): what `params()` below reads."""
        row = self.rows[index]
        cell = lambda v: _parse_cell(v) if isinstance(v, str) else v
        ctx = {"run_id": self.run_ids[index], "seed": int(self.seeds[index]), "db_path": None, "db_addr": None, "cache_dir": None,
               "run_dir": None if run_dir is None else str(run_dir),
               "params": {k[len("param."):]: cell(v) for k, v in row.items() if k.startswith("param.") and v != ""},
               "meta": {k[len("meta."):]: cell(v) for k, v in row.items() if k.startswith("meta.") and v != ""}, "slots": {}}
        ctx.update(fields)
        return ctx

    def to_csv(self) -> str:
        """Same text `python -m elodin.monte_carlo.sample` writes (csv.DictWriter defaults)."""
        import csv
        buf = io.StringIO(newline="")
        wr = csv.DictWriter(buf, fieldnames=self.headers)
        wr.writeheader()
        for rid, seed, row in zip(self.run_ids, self.seeds, self.rows):
            wr.writerow({"run_id": rid, "seed": int(seed), **row})
        return buf.getvalue()


def materialize(spec: Mapping[str, Any]) -> Plan:
    """spec = parsed spec.toml: optional [sim_sweep], [meta_sweep], [monte_carlo]{n_samples, seed, method, variables}."""
    mc = spec.get("monte_carlo")
    if not mc:
        mc_rows: List[Dict[str, Any]] = [{}]
    else:
        n = int(mc.get("n_samples", 1))
        if n < 1:
            raise ValueError(f"n_samples must be >= 1 (got {n})")
        method = str(mc.get("method", "lhs")).lower()
        if method not in ("lhs", "random"):
            raise ValueError(f'unknown method "{method}" (known: lhs, random)')
        rng = random.Random(mc.get("seed"))
        raw = dict(mc.get("variables", {}))
        variables = [Variable.parse(k, raw[k]) for k in raw]   # validate in declaration order
        variables.sort(key=lambda v: v.name)                   # sample in sorted-name order
        units = _unit_samples(n, len(variables), method, rng)
        mc_rows = [{f"param.{v.name}": v.at(u) for v, u in zip(variables, urow)} for urow in units]
    rows = []
    for sim_row, meta_row, mc_row in itertools.product(_grid(spec.get("sim_sweep"), "param"),
                                                       _grid(spec.get("meta_sweep"), "meta"), mc_rows):
        rows.append({**sim_row, **mc_row, **meta_row})
    headers = ["run_id", "seed"] + sorted({k for row in rows for k in row})
    return Plan(headers, rows, [f"run_{i:07d}" for i in range(len(rows))],
                np.arange(1, len(rows) + 1, dtype=np.uint64))


def load_spec(path) -> Dict[str, Any]:
    try:
        import tomllib
    except ModuleNotFoundError:  # Python 3.10
        import tomli as tomllib
    with open(path, "rb") as f:
        return tomllib.load(f)


# ---- the sim side of a campaign: el.monte_carlo.{Param, params_spec, params, result, port, spec_json} ------------------
# (libs/nox-py/src/monte_carlo.rs:12-331).  A sim script declares its parameters with defaults, asks `params()` for the
# values of THIS run — defaults overlaid with what the runner put in the context document — and reports scalars back with
# `result(...)`.  On this backend a whole campaign is one GPU job over `Plan.table()`; this API is what keeps a
# reference sim script importable and lets a single rollout be replayed from its context file.

CONTEXT_ENV = "ELODIN_MONTE_CARLO_CONTEXT"
_declared_spec: Optional["ParamsSpec"] = None


def _parse_cell(value: str):
    import json
    try:
        return json.loads(value)
    except ValueError:
        return value


def _jsonable(v):
    if v is None or isinstance(v, (bool, int, str)):
        return v
    if isinstance(v, float):
        if not np.isfinite(v):
            raise ValueError("float values must be finite")
        return v
    if isinstance(v, list):
        return [_jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    raise TypeError(f"value is not JSON serializable: {type(v).__name__}")


class Param:
    def __init__(self, type_, default=None, min=None, max=None):
        self.type_name = type_.__name__ if isinstance(type_, type) else (type_ if isinstance(type_, str) else repr(type_))
        self.default, self.min, self.max = _jsonable(default), _jsonable(min), _jsonable(max)

    def data(self) -> Dict[str, Any]:
        return {"type_name": self.type_name, "default": self.default, "min": self.min, "max": self.max}


class ParamsSpec:
    def __init__(self, params: Mapping[str, Param]):
        self.params = dict(params)

    def to_json(self) -> str:
        import json
        return json.dumps({"params": {k: v.data() for k, v in self.params.items()}}, indent=2)


def params_spec(**kwargs) -> ParamsSpec:
    for key, value in kwargs.items():
        if not isinstance(value, Param):
            raise TypeError(f"params_spec value for `{key}` must be el.monte_carlo.Param")
    global _declared_spec
    _declared_spec = ParamsSpec(kwargs)
    return _declared_spec


def spec_json() -> str:
    return (_declared_spec or ParamsSpec({})).to_json()


class Params:
    def __init__(self, values: Mapping[str, Any], ctx: Optional[Mapping[str, Any]] = None):
        ctx = ctx or {}
        self._params = dict(values)
        self.run_id, self.seed = ctx.get("run_id"), ctx.get("seed")
        self.db_path, self.db_addr = ctx.get("db_path"), ctx.get("db_addr")
        self.cache_dir, self.run_dir = ctx.get("cache_dir"), ctx.get("run_dir")
        self._meta, self._slots = dict(ctx.get("meta") or {}), dict(ctx.get("slots") or {})

    def get(self, key: str, default=None):
        return self._params.get(key, default)

    def __getitem__(self, key: str):
        return self._params[key]

    def as_overrides_dict(self) -> Dict[str, Any]:
        return dict(self._params)

    @property
    def meta(self) -> Dict[str, Any]:
        return dict(self._meta)

    def slots(self) -> Dict[str, Any]:
        return dict(self._slots)

    def ports(self) -> Dict[str, int]:
        ok = lambda v: isinstance(v, int) and not isinstance(v, bool) and 0 <= v <= 0xFFFF
        out = {k: v for k, v in (self._slots.get("ports") or {}).items() if ok(v)} if isinstance(self._slots.get("ports"), dict) else {}
        for name, v in self._slots.items():
            if name.endswith("_port") and ok(v):
                out.setdefault(name[:-len("_port")], v)
        return out


def params(spec: Optional[ParamsSpec] = None) -> Params:
    import json
    import os
    spec = spec or _declared_spec
    values = {k: p.default for k, p in spec.params.items()} if spec else {}
    path = os.environ.get(CONTEXT_ENV)
    if path is None:
        return Params(values)
    try:
        with open(path) as f:
            text = f.read()
    except OSError as err:
        raise RuntimeError(f"failed to read {CONTEXT_ENV}={path}: {err}") from None
    ctx = json.loads(text)
    values.update(ctx.get("params") or {})
    return Params(values, ctx)


def port(name: str, default: Optional[int] = None) -> int:
    import os
    env = "ELODIN_MC_PORT_" + "".join(ch.upper() if ch.isascii() and ch.isalnum() else "_" for ch in name)
    raw = os.environ.get(env)
    if raw is not None:
        try:
            v = int(raw)
            if not 0 <= v <= 0xFFFF:
                raise ValueError("out of range")
            return v
        except ValueError as err:
            raise ValueError(f"invalid port value for `{name}`: {err}") from None
    found = params(None).ports().get(name)
    if found is not None:
        return found
    if default is None:
        raise KeyError(name)
    return default


_active_result = [None]      # vectorize.Campaign: the record of the run whose callback is executing (all runs share the process)


def result(**kwargs) -> None:
    """Write the run's scalars to <run_dir>/result.json, where the campaign's post_run hook reads them.  Inside a vectorised
    campaign (all runs in one process, elodin_amd/vectorize.py) they land in that run's record instead."""
    import json
    from pathlib import Path
    if not kwargs:
        return
    if _active_result[0] is not None:
        _active_result[0].update({k: _jsonable(v) for k, v in kwargs.items()})
        return
    run_dir = params(None).run_dir
    if run_dir is None:
        raise RuntimeError("result() requires ELODIN_MONTE_CARLO_CONTEXT with run_dir")
    (Path(run_dir) / "result.json").write_text(json.dumps({k: _jsonable(v) for k, v in kwargs.items()}, indent=2))
