"""A Monte-Carlo campaign of a reference sim script as ONE executor: rows = runs.

The reference flies a campaign as one OS process per run (libs/monte-carlo/src/lib.rs:2083 `run_one`): each process calls
the script's `build(params)` with its own parameter values, which the script bakes into the traced code as Python floats
(`mass = float(params.get("mass", 1.5))`, examples/monte-carlo/sim.py:72-75) and into the spawned components
(`el.C(Velocity, jnp.array([wind]))`).  On a GPU the runs are rows of one world:

* the program is traced ONCE, with every parameter holding a distinctive sentinel value; a constant that equals a sentinel is
  that parameter and becomes a per-run column `mc:<name>` (elodin_amd.dsl.PARAM_SENTINELS).  Tracing twice with two sentinel
  sets and comparing the generated sources proves that no parameter reached the code in a form the tracer could not see
  (host arithmetic on a parameter before the traced function — `k = 2.0 * mass` outside it — would differ between the two
  and is refused: such a script needs one program per run);
* the spawned components are read from `build(params_i)` run on the host for every run (its traced function is never
  called), entity e of run i becoming row i * E + e;
* `post_step(tick, ctx)` callbacks — how the reference's scripts exchange with their controllers — are called per run with a
  StepContext that sees that run's entities under their own names, on the server loop's cadence (compat.run_stepwise);
  `el.monte_carlo.result(...)` lands in the run's result record instead of <run_dir>/result.json.

Used by tests/test_gpu_monte_carlo_example.py and bench.py's `monte_carlo_example` leg on examples/monte-carlo.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence

import numpy as np

from . import api as _api
from . import dsl as _dsl
from . import monte_carlo as _mc


def _sentinels(names: Sequence[str], spec: Optional[_mc.ParamsSpec], variant: int) -> Dict[str, float]:
    """One distinctive value per parameter inside its declared [min, max]: irrational-looking fractions no literal of a
    script will equal, different for the two variants."""
    out = {}
    for k, name in enumerate(names):
        p = spec.params.get(name) if spec is not None else None
        lo = float(p.min) if p is not None and p.min is not None else 0.5
        hi = float(p.max) if p is not None and p.max is not None else 1.5
        frac = ((k + 1) * 0.6180339887498949 + (0.2718281828459045 if variant else 0.1414213562373095)) % 1.0
        out[name] = lo + (hi - lo) * (0.05 + 0.9 * frac)
    if len(set(out.values())) != len(out):
        raise RuntimeError("sentinel values collide")
    return out


class _sentinel_scope:
    def __init__(self, values: Mapping[str, float]): self.map = {float(v): k for k, v in values.items()}
    def __enter__(self):
        self.saved = dict(_dsl.PARAM_SENTINELS)
        _dsl.PARAM_SENTINELS.clear()
        _dsl.PARAM_SENTINELS.update(self.map)
    def __exit__(self, *a):
        _dsl.PARAM_SENTINELS.clear()
        _dsl.PARAM_SENTINELS.update(self.saved)


def _entities(world: _api.World) -> Dict[int, Dict[str, np.ndarray]]:
    """{entity id: {component: row}} of a spawned world, ids ascending."""
    out: Dict[int, Dict[str, np.ndarray]] = {}
    for cname in world._components:
        rows, ids = world.column(cname)
        for row, eid in zip(rows, ids):
            out.setdefault(int(eid), {})[cname] = row
    return dict(sorted(out.items()))


def plan_of(param_rows: Sequence[Mapping[str, float]]) -> _mc.Plan:
    """A plan from explicit parameter rows (run ids / seeds numbered like sample.py:149 does: run_%07d, seed = index)."""
    names = sorted({k for r in param_rows for k in r})
    return _mc.Plan(headers=["run_id", "seed"] + [f"param.{n}" for n in names],
                    rows=[{f"param.{n}": r[n] for n in names if n in r} for r in param_rows],
                    run_ids=[f"run_{i:07d}" for i in range(len(param_rows))], seeds=np.arange(len(param_rows), dtype=np.uint64))


class Campaign:
    """`build(params) -> (world, system)` of a sim script over the rows of a plan, as one executor.

    plan: monte_carlo.Plan; spec: the script's el.monte_carlo.params_spec (defaults / bounds); world_cls: the World class the
    script used (elodin_amd.frontend.World, or compat's el.World)."""

    def __init__(self, build: Callable, plan: _mc.Plan, spec: Optional[_mc.ParamsSpec] = None, *, simulation_rate: float = 120.0,
                 telemetry_rate: Optional[float] = None, device: int = 0, world_cls=None, dry: bool = False):
        self.plan, self.spec = plan, spec
        self.names = list(plan.param_names)
        defaults = {k: p.default for k, p in spec.params.items()} if spec is not None else {}
        self.table = plan.table(self.names, defaults)
        self.n_runs = len(plan)
        self.simulation_rate, self.telemetry_rate = float(simulation_rate), telemetry_rate
        if self.n_runs == 0:
            raise ValueError("an empty plan")

        def built(values, ctx=None):
            w, system = build(_mc.Params({**defaults, **values}, ctx))
            return w, system

        # -- the program: traced under two sentinel sets, the sources must agree ------------------------------------------
        sources, sentinels, worlds = [], [], []
        for variant in (0, 1):
            sent = _sentinels(self.names, spec, variant)
            sentinels.append(sent)
            w, system = built(sent)
            worlds.append((w, _entities(w)))
            for eid in _entities(w):
                w.insert(_api.EntityId(eid), [_api.C("mc:" + n, [sent[n]]) for n in self.names])
            with _sentinel_scope(sent):
                sources.append(w.generated_sources(system, simulation_rate=simulation_rate))
            if variant == 0:
                self._system, self._sent = system, sent
        if sources[0] != sources[1]:
            raise NotImplementedError(
                "this script's code depends on a Monte-Carlo parameter in a way the tracer cannot see (host arithmetic on the "
                "parameter before the traced function): the campaign cannot share one program — build one executor per run")
        self.sources = sources[0]
        self._sentinel_worlds = worlds
        # Two mid-range sentinels cannot see host CONTROL FLOW or truncation on a parameter (`if wind > 0:`, `int(p)`,
        # `range(int(p))`, a table row picked by a parameter): both take the same path and generate the same text.  So the
        # same comparison is repeated at both ends of every parameter's range and on real rows of the plan (each value nudged
        # by a relative 1e-12-scale irrational so no literal of the script equals it): a script whose host code branches on a
        # parameter generates different text somewhere along the way and is refused like host arithmetic is.
        self._probe_values = self._probes(self.names, spec, self.table)
        for values in self._probe_values:
            w, system = built(values)
            for eid in _entities(w):
                w.insert(_api.EntityId(eid), [_api.C("mc:" + n, [values[n]]) for n in self.names])
            with _sentinel_scope(values):
                probe_src = w.generated_sources(system, simulation_rate=simulation_rate)
            if probe_src != self.sources:
                raise NotImplementedError(
                    "this script's host code takes a different path for different values of a Monte-Carlo parameter (a Python `if`, "
                    "`int()`, `round()`, `range()` or a table lookup on the parameter): the traced program differs between runs, so "
                    f"the campaign cannot share one program — build one executor per run (probe values: {values})")

        # -- the world: every run's spawned entities, run-major ------------------------------------------------------------
        # A spawned value that is the same under both sentinel sets is a constant of the script; one that equals a sentinel
        # under both is that parameter (`el.C(Velocity, jnp.array([wind]))`): such worlds are laid out from the plan table
        # directly.  Any other dependence on the parameters (host arithmetic before the spawn) needs build(params_i) per run.
        from . import frontend as _fe
        self.world = (world_cls or _fe.World)()
        self.entity_names: List[Dict[str, int]] = []          # per run: the script's entity name -> entity id in the big world
        ents_a, ents_b = self._sentinel_worlds
        template = self._spawn_template(ents_a, ents_b, sentinels)
        if template is not None:      # the same probes for the spawned values: a spawn that branches on a parameter needs build(params_i)
            for values in self._probe_values:
                w, _ = built(values)
                ents = _entities(w)
                if list(ents) != list(template) or any(
                        set(ents[eid]) != set(comps) or any(not np.array_equal(np.asarray(ents[eid][c], dtype=np.float64),
                                                            np.asarray(row if src is None else self._fill(row, src, values), dtype=np.float64))
                                                            for c, (row, src) in comps.items())
                        for eid, comps in template.items()):
                    template = None
                    break
        self.per_run_builds = template is None
        self.entities_per_run = len(ents_a[1])
        w0 = ents_a[0]
        for i in range(self.n_runs):
            values = {n: float(self.table[i, k]) for k, n in enumerate(self.names)}
            if template is None:
                w, _ = built(values, plan.context(i))
                ents = _entities(w)
                if len(ents) != self.entities_per_run:
                    raise NotImplementedError("runs of one campaign spawn different numbers of entities")
            else:
                w, ents = w0, {eid: {c: (row if src is None else self._fill(row, src, values)) for c, (row, src) in comps.items()}
                               for eid, comps in template.items()}
            local_to_big, names = {}, {}
            for eid, comps in ents.items():
                arch = [_api.C(c, row) for c, row in comps.items() if not c.startswith("mc:")] + [_api.C("mc:" + n, [values[n]]) for n in self.names]
                big = self.world.spawn(arch, name=f"{plan.run_ids[i]}.{w._names.get(eid, eid)}")
                local_to_big[eid] = int(big)
                if eid in w._names:
                    names[w._names[eid]] = int(big)
            for nm, eid in w.entity_ids_by_name.items():
                names[nm] = local_to_big[eid]
            for comp, pairs in w._edges.items():
                for a, b in pairs:
                    self.world.insert(_api.EntityId(local_to_big[a]), [_api.GravityEdge(local_to_big[a], local_to_big[b], comp)])
            self.entity_names.append(names)
        self.exec = None
        if not dry:
            with _sentinel_scope(self._sent):
                kw = {"history": False} if "history" in self.world.build.__code__.co_varnames else {}
                self.exec = self.world.build(self._system, simulation_rate=simulation_rate, telemetry_rate=telemetry_rate,
                                             device=device, **kw)
        self.results: List[Dict[str, Any]] = [dict() for _ in range(self.n_runs)]

    @staticmethod
    def _probes(names, spec, table, max_rows: int = 6):
        """Parameter value sets at which the program is traced again: both ends of every range (the spec's bounds, else the plan
        table's extremes) and up to `max_rows` real plan rows spread over the plan, every value nudged to be distinctive."""
        def nudge(v, k, salt):
            eps = (1.0 + ((k + 1) * 0.6180339887498949 + salt) % 1.0) * 1e-12
            return float(v) + (abs(float(v)) if v else 1.0) * eps
        out = []
        if not names:
            return out
        lo_hi = []
        for k, name in enumerate(names):
            p = spec.params.get(name) if spec is not None else None
            col = table[:, k]
            lo = float(p.min) if p is not None and p.min is not None else float(np.min(col))
            hi = float(p.max) if p is not None and p.max is not None else float(np.max(col))
            lo_hi.append((lo, hi))
        if any(lo != hi for lo, hi in lo_hi):
            out.append({n: nudge(lo, k, 0.31) for k, (n, (lo, _)) in enumerate(zip(names, lo_hi))})
            out.append({n: nudge(hi, k, 0.73) - 2e-12 * (abs(hi) if hi else 1.0) * 2 for k, (n, (_, hi)) in enumerate(zip(names, lo_hi))})
        rows = sorted(set(int(round(x)) for x in np.linspace(0, len(table) - 1, min(max_rows, len(table)))))
        for j, i in enumerate(rows):
            out.append({n: nudge(table[i, k], k, 0.11 * (j + 1)) for k, n in enumerate(names)})
        good = []
        for values in out:      # distinct values per set (the sentinel map is keyed on them)
            if len(set(values.values())) == len(values):
                good.append(values)
        return good

    @staticmethod
    def _spawn_template(a, b, sentinels):
        """{entity: {component: (row, None | [parameter name or None per element])}} when every spawned element is a constant
        or exactly one of the parameters; None when a run's spawn cannot be derived from the plan row that simply."""
        (wa, ea), (wb, eb) = a, b
        if list(ea) != list(eb) or wa._names != wb._names or wa._edges != wb._edges:
            return None
        inv_a = {v: k for k, v in sentinels[0].items()}
        inv_b = {v: k for k, v in sentinels[1].items()}
        out = {}
        for eid in ea:
            if set(ea[eid]) != set(eb[eid]):
                return None
            out[eid] = {}
            for c, ra in ea[eid].items():
                rb = eb[eid][c]
                if ra.shape != rb.shape:
                    return None
                if np.array_equal(ra, rb):
                    out[eid][c] = (ra, None)
                    continue
                src = []
                for x, y in zip(ra.reshape(-1), rb.reshape(-1)):
                    if x == y:
                        src.append(None)
                    elif inv_a.get(float(x)) is not None and inv_a.get(float(x)) == inv_b.get(float(y)):
                        src.append(inv_a[float(x)])
                    else:
                        return None
                out[eid][c] = (ra, src)
        return out

    @staticmethod
    def _fill(row, src, values):
        out = np.array(row, dtype=np.float64).reshape(-1)
        for k, name in enumerate(src):
            if name is not None:
                out[k] = values[name]
        return out.reshape(row.shape)

    # ---------------------------------------------------------------------------------------------------------------------
    def run(self, max_ticks: int, post_step: Optional[Callable] = None, pre_step: Optional[Callable] = None) -> "Campaign":
        """The server loop over all runs in lock-step.  Callbacks are the script's own, called once per run and batch with that
        run's StepContext (a write is uploaded before the next batch).  Without callbacks: one launch sequence, no host work."""
        ex = self.exec
        if post_step is None and pre_step is None:
            ex.run(int(max_ticks))
            return self
        from . import compat as _compat
        tpt = max(1, int(round(self.simulation_rate / self.telemetry_rate))) if self.telemetry_rate else 1
        ctxs = []
        for i in range(self.n_runs):
            c = _compat.StepContext(ex, self.world, 1.0 / self.simulation_rate)
            c._entity = dict(self.entity_names[i])
            c._entity.update({_compat._snake(k): v for k, v in self.entity_names[i].items()})
            c.run_index, c.run_id = i, self.plan.run_ids[i]          # beyond the reference's StepContext: which run this is
            ctxs.append(c)
        tick, limit = ex.tick, int(max_ticks)
        saved = _mc._active_result[0]
        try:
            while tick < limit:
                batch = max(1, min(tpt, limit - tick))
                if pre_step is not None:
                    for i, c in enumerate(ctxs):
                        c._tick = tick
                        _mc._active_result[0] = self.results[i]
                        pre_step(tick, c)
                if any(c._dirty for c in ctxs):
                    ex._hip.upload()
                    for c in ctxs:
                        c._dirty = False
                ex.run(batch)
                tick = ex.tick
                if post_step is not None:
                    for i, c in enumerate(ctxs):
                        c._tick = tick - 1
                        _mc._active_result[0] = self.results[i]
                        post_step(tick - 1, c)
            if any(c._dirty for c in ctxs):
                ex._hip.upload()
        finally:
            _mc._active_result[0] = saved
        return self

    def traced(self):
        """(TracedProgram, {column: initial rows}, dt) of the campaign's program without a device: what a CPU walk of the
        trace needs (tests/dsl_numpy.py)."""
        with _sentinel_scope(self._sent):
            plan = self.world.build(self._system, simulation_rate=self.simulation_rate, telemetry_rate=self.telemetry_rate, _dry=True)
            tp = plan["effectors"].trace()
        return tp, {n: np.asarray(plan["columns"][n], dtype=np.float64) for n, _ in tp.columns}, plan["dt"]

    def column(self, component: str) -> np.ndarray:
        """[n_runs * entities carrying it, w] rows of a component, run-major."""
        return np.asarray(self.exec.column_array(component))

    def result_table(self, names: Sequence[str]) -> np.ndarray:
        """The runs' `el.monte_carlo.result(...)` records as [n_runs, len(names)] (NaN where a run reported nothing)."""
        out = np.full((self.n_runs, len(names)), np.nan)
        for i, rec in enumerate(self.results):
            for k, n in enumerate(names):
                if n in rec:
                    out[i, k] = float(rec[n])
        return out
