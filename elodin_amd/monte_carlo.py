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
from typing import Any, Dict, List, Mapping, Sequence

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
